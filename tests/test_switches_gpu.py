"""Every KAI0_* switch that selects another kernel, schedule or cache policy is a configuration of the shipped library
(VERDICT r2, weak #13): each one runs the full-width one-layer model (tests/switch_probe.py, a subprocess per setting: the library
reads its switches once per process) and must reproduce the default's training loss, gradients and action chunk — bit for bit
where the switch only moves data (cache hints, grid order, where codes are read from), within the bf16 path's round-off where it
changes the order of a sum (another kernel for the same op)."""

import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

# (environment, must be bit-identical)
SWITCHES = [
    ({"KAI0_GEMM_NT": "0"}, True),                              # plain instead of non-temporal stores
    ({"KAI0_INFER_GRAPH": "0"}, True),                          # eager launches instead of the hipGraph replay
    ({"KAI0_GEGLU_PAIR": "0"}, False),                          # gate GEMM + up GEMM (act 2) instead of the pair GEMM (the GEMMs are
                                                                # bit-identical; the backward's operand layout differs)
    ({"KAI0_INFER_CACHE_MODS": "0"}, False),                    # modulation table recomputed per call (hence no folded adaRMS weights)
    ({"KAI0_INFER_FOLD": "0", "KAI0_ATTN_STORE_P": "1", "KAI0_GEMM_PERSIST": "0"}, False),  # round 3's forms: adaRMS prologue in the
                                                                # denoise kernels, stored-P attention, one GEMM block per tile
    ({"KAI0_GEMM_PERSIST": "2"}, True),                         # every eligible NT GEMM on the persistent kernel
    ({"KAI0_SKIP_DEAD_PREFIX": "0"}, True),                     # the last layer's dead prefix o_proj / MLP computed
    ({"KAI0_ZERO_GRADS": "full"}, True),                        # flat gradient buffers cleared every step
    ({"KAI0_DEFER_REDUCE": "0"}, True),                         # norm-weight / bias gradient sums launched one by one, not queued
    ({"KAI0_SPARSE_EMBED": "0"}, True),                         # embedding table through the dense AdamW pass (no idle-row skip)
    ({"KAI0_EXPERT_STREAM": "0"}, True),                        # action expert's chain on the main stream
    ({"KAI0_SK2_PACKED": "0"}, True),                           # denoise kernels: row-major instead of fragment-major weights
    ({"KAI0_INFER_FUSE_NORM": "0", "KAI0_PREFIX_SPLITS": "1,1,6"}, False),  # norms as launches of their own, unsplit o_proj
    ({"KAI0_INFER_GLUE": "0"}, False),                          # six-launch step seam
    ({"KAI0_FUSE_QKV": "0", "KAI0_SIGLIP_BWD": "gemm", "KAI0_SIGLIP_FWD": "general"}, False),  # three projection GEMMs; SigLIP attention
                                                                # on the general forward kernel + the GEMM-based backward
    ({"KAI0_INFER_INBLOCK": "0"}, False),                       # split-K denoise GEMMs + combine launches
]


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-12))


def _run(tmp_path, env, tag):
    out = tmp_path / f"{tag}.pt"
    e = {k: v for k, v in os.environ.items() if not k.startswith("KAI0_") or k in ("KAI0_HIP_LIB",)}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(HERE, "switch_probe.py"), str(out)], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{env}: probe failed\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}"
    return torch.load(out, weights_only=True)


@pytest.fixture(scope="module")
def default_run(tmp_path_factory):
    return _run(tmp_path_factory.mktemp("switches"), {}, "default")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("env,exact", SWITCHES, ids=[" ".join(f"{k}={v}" for k, v in e.items()) for e, _ in SWITCHES])
def test_switch_reproduces_the_default(default_run, tmp_path, env, exact):
    got = _run(tmp_path, env, "variant")
    assert set(got) == set(default_run)
    worst = max(rel(got[k], default_run[k]) for k in got)
    same = all(torch.equal(got[k], default_run[k]) for k in got)
    print(f"{env}: bit-identical={same} worst rel-L2 {worst:.3e}")
    if exact:
        assert same, {k: rel(got[k], default_run[k]) for k in got if not torch.equal(got[k], default_run[k])}
    else:
        assert rel(got["loss"], default_run["loss"]) < 5e-3 and rel(got["chunk"], default_run["chunk"]) < 3e-3
        assert all(rel(got[k], default_run[k]) < 3e-2 for k in got if k.startswith("grad.")), {k: rel(got[k], default_run[k]) for k in got}
