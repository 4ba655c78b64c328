"""Build libvist3a_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the library is a plain
C-ABI shared object (include/vist3a_hip.h) loaded through ctypes by vist3a_amd.lib."""
from __future__ import annotations

import hashlib
import json
import os
import re
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
# hipcc 7.2 is fragile around a few of the attention instantiations: a harmless source change once made two of them spill 3.3 KB per lane
# (a 21 us launch became 950 us; every test still passed).  The compiler's resource remarks are therefore part of the build: a kernel that
# needs more scratch than this fails it.  (Largest in the shipped library: 116 bytes per lane.)
MAX_SCRATCH_BYTES_PER_LANE = int(os.environ.get("V3A_MAX_SCRATCH", "512"))


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
    cmd = [_hipcc(), *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", str(src), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stderr}")
    usage = kernel_resources(r.stderr)
    (OBJ / (src.stem + ".resources.json")).write_text(json.dumps(usage, indent=1))
    heavy = {k: v["scratch"] for k, v in usage.items() if v["scratch"] > MAX_SCRATCH_BYTES_PER_LANE}
    if heavy:
        obj.unlink(missing_ok=True)
        raise RuntimeError(f"{src.name}: kernels spilling more than {MAX_SCRATCH_BYTES_PER_LANE} bytes per lane (V3A_MAX_SCRATCH overrides): {heavy}")
    stamp.write_text(want)
    return obj, True


def kernel_resources(remarks: str) -> dict:
    """{mangled kernel name: {vgprs, scratch (bytes per lane), occupancy (waves per SIMD)}} from hipcc's kernel-resource-usage remarks."""
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark: (?:\s*)(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "Function Name":
            cur = out.setdefault(val, {"vgprs": 0, "scratch": 0, "occupancy": 0})
        elif cur is not None:
            cur[{"VGPRs": "vgprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy"}[key]] = int(val)
    return out


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
