"""Hardware parity evidence for the OTHER 16-bit element type: a process loads ONE of the two libraries, so
the parity subset — reference golden, per-block taps against the oracle, the fused CFG + DDIM loop, the GEMM / attention /
row-stationary kernel tests — is run once more in a child process with VMV_DTYPE=bf16 (the tests switch to the bf16 tolerances of
DESIGN.md §6 by the loaded library: 3e-2 / 3e-2 / 6e-2; bf16 is the range fallback, fp16 the default)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SUBSET = ("test_unet_matches_reference_golden or test_unet_blocks_match_oracle or test_fused_cfg_ddim_loop_matches_oracle or "
          "test_full_size_architecture_parity_small_latent or test_gemm_linear_bias or test_gemm_geglu or test_gemm_rs_ or "
          "test_gemm_conv3x3 or test_gemm_temporal_conv or test_gemm_tfr or test_gemm_tqa or test_gemm_layernorm_folded or test_attention or test_groupnorm or "
          "test_layernorm")


def test_other_element_type_parity_in_child_process():
    if os.environ.get("VMV_DTYPE_CHILD"):
        pytest.skip("already the child run")
    from videomv_amd import _lib as L
    other = "bf16" if L.elem_name() == "fp16" else "fp16"
    env = dict(os.environ, VMV_DTYPE=other, VMV_DTYPE_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_unet_gpu.py", "tests/test_kernels_gpu.py", "-q", "-x", "-m", "gpu",
                        "-p", "no:cacheprovider", "-k", SUBSET], cwd=ROOT, env=env, capture_output=True, text=True, timeout=2400)
    tail = r.stdout[-4000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail
    print(f"[{other}] " + r.stdout.strip().splitlines()[-1])
