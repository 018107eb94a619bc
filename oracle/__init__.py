"""CPU oracle for the medimgen hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU (fp32) restatement of the two network files
of the reference (`medimgen/diffusion_model_unet_with_strides.py`,
`medimgen/autoencoderkl_with_strides.py`) and of the train-step semantics of
`medimgen/train_ldm.py` / `train_ddpm.py` / `train_autoencoder.py`.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  The product package
(`medical_image_generation_amd/`) never does: its ops fail loudly when the HIP
library is missing instead of falling back to anything in here.

Pinning: `oracle/tools/gen_golden.py` runs the reference's own two model files
(in the build container only; they need a 4-symbol stand-in for `monai`, see
`oracle/tools/monai_standin.py`) and writes the fixtures under `tests/golden/`;
`tests/test_oracle_golden.py` checks this restatement against them.
"""
