"""Regenerate the full-size oracle digests (tests/golden/oracle_*.safetensors) on HOST CORES ONLY: no GPU, no HIP library - the cases of
tests/fullsize_cases.py are run by the CPU oracle and the sampled outputs written with their fingerprint (inputs + weights checksum and the
sha256 of the oracle sources).  ~20 minutes on the GPU box's 128 host threads (hours on 8):

    gpurun --timeout 2400 -- 'HIP_VISIBLE_DEVICES= python tests/golden/make_fullsize_oracle.py --out gpurun_out/golden'
    cp gpurun_out/golden/oracle_*.safetensors tests/golden/

    python tests/golden/make_fullsize_oracle.py [--out DIR] [case ...]        cases: recon_full recon_config3 dit_full_depth"""
import argparse
import os
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parents[1]))
sys.path.insert(0, str(HERE.parent))

import torch  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(HERE))
    ap.add_argument("cases", nargs="*")
    a = ap.parse_args()
    os.environ["V3A_LIVE_ORACLE"], os.environ["V3A_WRITE_ORACLE"], os.environ["V3A_ORACLE_OUT"] = "1", "1", a.out
    import fullsize_cases as FC
    import oracle_cache as OC
    assert not torch.cuda.is_available() or os.environ.get("HIP_VISIBLE_DEVICES") == "", "run with HIP_VISIBLE_DEVICES= : this generator must not depend on a GPU"
    for name in a.cases or list(FC.CASES):
        t0 = time.time()
        c = FC.CASES[name]()
        _, live = OC.oracle(c.name, c.fingerprint, c.compute, sources=c.sources, case_fns=c.case_fns)
        assert live
        print(f"{name}: oracle_{c.name}.safetensors written to {a.out} in {time.time() - t0:.0f} s ({torch.get_num_threads()} threads)", flush=True)
