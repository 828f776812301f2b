import json
import sys
from pathlib import Path

import pytest
import torch

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / "tests" / "golden"
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- parity record: every figure a GPU test measures against the reference's tensors (bit-exact fractions, ulp
# maxima, relative RMS vs the fp32 yardstick, greedy-id prefix lengths) is collected here and written to
# $DUALHYP_PARITY_JSON (default gpurun_out/parity.json) when the session ends; the committed copy is profiles/rNN_parity.json
_PARITY = {}


def record_parity(key: str, **figures) -> None:
    _PARITY[key] = {k: (float(v) if isinstance(v, (int, float)) or hasattr(v, "__float__") else v) for k, v in figures.items()}
    print(f"[parity] {key}: " + " ".join(f"{k}={v}" for k, v in _PARITY[key].items()))


def pytest_sessionfinish(session, exitstatus):
    import os
    if not _PARITY:
        return
    path = Path(os.environ.get("DUALHYP_PARITY_JSON", REPO / "gpurun_out" / "parity.json"))
    try:
        path.parent.mkdir(parents=True, exist_ok=True)
        old = json.loads(path.read_text()) if path.exists() else {}
        old.update(_PARITY)
        path.write_text(json.dumps(old, indent=1, sort_keys=True))
    except OSError:
        pass


def load_golden(name: str):
    """-> (dict of tensors, meta dict) from tests/golden/<name>.safetensors."""
    from safetensors import safe_open
    f = safe_open(str(GOLDEN / f"{name}.safetensors"), "pt")
    tensors = {k: f.get_tensor(k) for k in f.keys()}
    meta = json.loads(f.metadata().get("meta", "{}"))
    return tensors, meta


def bf16_ulp(x: torch.Tensor) -> torch.Tensor:
    """Spacing of bf16 numbers at |x| (8 significant bits)."""
    a = x.float().abs().clamp_min(2.0 ** -126)
    return torch.exp2(torch.floor(torch.log2(a)) - 7)


def ulp_diff(a: torch.Tensor, b: torch.Tensor, floor_frac: float = 1.0 / 64) -> torch.Tensor:
    """|a-b| in units of the bf16 ulp at max(|a|, |b|, floor) with floor = floor_frac * rms(b):
    a dot product that cancels to ~0 carries the absolute error of its O(rms) partial sums, so
    its own (tiny) ulp is not the right yardstick."""
    a, b = a.float(), b.float()
    floor = floor_frac * b.pow(2).mean().sqrt().clamp_min(1e-30)
    return (a - b).abs() / bf16_ulp(torch.maximum(torch.maximum(a.abs(), b.abs()), floor))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get
