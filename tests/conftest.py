import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def hip_lib():
    """The C-ABI library, built in-tree; GPU tests must run THROUGH it (no fallback)."""
    from vist3a_amd import lib
    return lib.load()


@pytest.fixture(scope="session")
def recon_full(hip_lib):
    """Width-1024 / 16-head reconstruction weights (tests/fullsize_cases.py::recon_full_weights): (oracle ReconCfg, state dict).  Built once
    per session - the seeded fp32 generator takes ~20 s of host time - and shared by the full-size and the per-block production tests."""
    sys.path.insert(0, str(ROOT / "tests"))
    import fullsize_cases as FC
    return FC.recon_full_weights()


class _ParityLog:
    """Measured parity errors of the `-m gpu` tests, written to gpurun_out/parity.json at session end (the builder copies the
    file to profiles/rNN/parity.json so that every tolerance quoted in DESIGN.md has an artifact behind it)."""

    def __init__(self):
        self.rows = []

    def __call__(self, test: str, **kw):
        self.rows.append(dict(test=test, **{k: (float(v) if hasattr(v, "__float__") and not isinstance(v, (bool, int, str)) else v)
                                            for k, v in kw.items()}))


_PARITY = _ParityLog()


@pytest.fixture(scope="session")
def parity():
    return _PARITY


def pytest_sessionfinish(session, exitstatus):
    import json
    out = Path(os.environ.get("V3A_PARITY_JSON", ROOT / "gpurun_out" / "parity.json"))
    worker = os.environ.get("PYTEST_XDIST_WORKER")
    try:
        if worker:   # an xdist worker: leave its rows for the controller (whose sessionfinish runs after every worker's)
            if _PARITY.rows:
                out.parent.mkdir(parents=True, exist_ok=True)
                out.with_name(f"{out.stem}.{worker}.part").write_text(json.dumps(_PARITY.rows))
            return
        rows = list(_PARITY.rows)
        for part in sorted(out.parent.glob(f"{out.stem}.*.part")):
            rows += json.loads(part.read_text())
            part.unlink()
        if not rows:
            return
        out.parent.mkdir(parents=True, exist_ok=True)
        meta = {}
        try:
            import torch
            meta = dict(device=torch.cuda.get_device_name(0), torch=torch.__version__)
        except Exception:  # noqa: BLE001
            pass
        out.write_text(json.dumps(dict(meta=meta, rows=sorted(rows, key=lambda r: r.get("test", ""))), indent=1))
    except OSError:
        pass
