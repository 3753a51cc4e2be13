"""CPU oracle for the UCOD-DPL data-parallel hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ucod_dpl_amd/`` may import this
package.  The only permitted importers are ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` -- and there only as the checker,
never as the thing measured or shipped.

What it is: a plain restatement, with torch-CPU tensors used as an array library
(fp32 by default, fp64 on request), of the arithmetic the reference performs on
the path named by BASELINE.json:north_star.  Each function cites the reference
``file:line`` it follows (paths relative to /root/reference).

Pinning: the reference ships no tests, so every function here
is pinned against outputs of the *imported reference itself*, captured in this
container by ``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``
(see ``tests/test_oracle_golden.py``).  The ViT backbone arithmetic lives in an
un-vendored dependency (HuggingFace ``transformers``, unpinned in the reference's
requirement.txt:6; 5.15.0 installed here): ``oracle/vit.py`` restates
``transformers/models/dinov2/modeling_dinov2.py`` of that version and the in-repo
``models/backbones/dino.py`` and is pinned by fixtures generated from both.
The LoRA backbone-backward mode (models/modules/full_model.py) is unimportable in
the reference and therefore: parity unpinned (not built this round).
"""

from . import decoder, discriminator, apm, train_step, look_twice, vit, resize  # noqa: F401
