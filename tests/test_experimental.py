"""Experiment records: kernel organisations that were built, measured and REJECTED on speed (LABNOTES R5.1, R5.2, R5.4, R5.6, R6.1).
They are not in the default library; `tools/build_variant.sh exp -DSAIS_EXPERIMENTAL=1` builds tools/bin/exp/libsais_hip.so with
them, and this file holds them to the same parity tests as the shipped kernels.  Marked `experimental`, NOT `gpu`: the driver's
`-m gpu` run does not include them (they need a GPU and the experimental library, and skip without either)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.experimental
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_LIB = os.path.join(ROOT, "tools", "bin", "exp", "libsais_hip.so")


@pytest.mark.parametrize("switch,value,select", [
    ("SAIS_NT_W4", "1", "gemm_nt_epilogues or gemm_nt_exact"),                  # four workgroups per CU, BK = 32 (R5.1)
    ("SAIS_NT_W16", "1", "gemm_nt_epilogues or gemm_nt_exact or gemm_patch"),   # two groups in anti-phase (R5.2)
    ("SAIS_NT_W8R", "1", "gemm_nt_epilogues or gemm_nt_exact"),                 # W in registers (R5.6)
    ("SAIS_ATTN_BWD_NB", "1", "vit_attention"),                                 # attention backward without the barrier chain (R6.5)
    ("SAIS_TN_XL", "8", "gemm_tn"),                                             # 192 x 384 dW tile with eight waves (R6.1)
    ("SAIS_TN_NI", "2", "gemm_tn")])                                            # 128 x 384 dW, two barrier intervals per step (R5.4)
def test_rejected_kernel_forms_pass_the_same_tests(switch, value, select):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(EXP_LIB):
        pytest.skip("no experimental library: tools/build_variant.sh exp -DSAIS_EXPERIMENTAL=1")
    env = dict(os.environ, SAIS_HIP_LIB=EXP_LIB, **{switch: value})
    if switch == "SAIS_TN_NI":
        env["SAIS_TN_XL"] = "0"
    sel = f"({select}) and not alternate and not slab_mode"
    if switch != "SAIS_TN_XL":
        sel += " and not bitwise_repeatable"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_kernels_gpu.py"), "-q", "-x", "-m", "gpu",
                        "-k", sel], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]
