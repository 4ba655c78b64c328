"""Build libvist3a_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a plain
C-ABI shared object (include/vist3a_hip.h) loaded through ctypes by vist3a_amd.lib."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = CSRC / "build"
LIB = HERE / "libvist3a_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"] + os.environ.get("V3A_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def _stamp(src: Path) -> str:
    h = hashlib.sha1()
    h.update(src.read_bytes())
    for dep in sorted(CSRC.glob("*.h")) + [HERE.parent / "include" / "vist3a_hip.h"]:
        h.update(dep.read_bytes())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: Path, force: bool) -> tuple[Path, bool]:
    OBJ.mkdir(exist_ok=True)
    obj = OBJ / (src.stem + ".o")
    stamp = OBJ / (src.stem + ".stamp")
    want = _stamp(src)
    if not force and obj.exists() and stamp.exists() and stamp.read_text() == want:
        return obj, False
    cmd = [_hipcc(), *FLAGS, "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
    stamp.write_text(want)
    return obj, True


def build(force: bool = False, verbose: bool = True) -> Path:
    srcs = sorted(CSRC.glob("*.hip"))
    if not srcs:
        raise RuntimeError(f"no .hip sources under {CSRC}")
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not LIB.exists():
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[vist3a_amd.build] linked {LIB} from {len(objs)} objects", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
