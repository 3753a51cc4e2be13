from .uscod import baseline  # noqa: F401
from .discriminator import Discriminator  # noqa: F401
from .UDLR import SparseRefiner  # noqa: F401
