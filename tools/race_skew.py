#!/usr/bin/env python3
"""Schedule-skew race check for the hand-written kernels.

    python tools/race_skew.py build              # here: srfrd_amd/lib/skew/libsrfrd_hip_<mask>.so for every mask
    python tools/race_skew.py run [pytest args]  # on the GPU box: the -m gpu suite once per skewed library

Each library is the product sources compiled with -DSRFRD_SKEW=<mask> (srfrd_dev.h): after every workgroup barrier the
waves of the mask sleep ~2 us, so the others run far ahead inside the barrier interval.  Correct kernels give the same
results under every mask; an unsynchronised read-early / write-late pair inside one interval fails the parity tests.
(The first-generation backward's end-of-block buffer swap was such a pair: it showed up once in some hundred full-suite
runs; under mask 0x0f0f it fails 43 of the 58 training tests every time.)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MASKS = ["0x0f0f", "0xf0f0", "0x5555", "0xaaaa", "0x3333", "0xcccc", "0x0001", "0xfffe", "0x00ff", "0xff00"]
OUT = os.path.join(ROOT, "srfrd_amd", "lib", "skew")


def build():
    import __graft_entry__ as g
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT, exist_ok=True)
    newest = max(os.path.getmtime(os.path.join(g.CSRC, f)) for f in g.SOURCES + g.HEADERS)
    for m in MASKS:
        lib = os.path.join(OUT, f"libsrfrd_hip_{m}.so")
        if os.path.exists(lib) and os.path.getmtime(lib) > newest:
            continue
        objdir = os.path.join(OUT, "obj_" + m)
        os.makedirs(objdir, exist_ok=True)
        procs = []
        for src in g.SOURCES:
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            extra = g.ENCODER_FLAGS if src.startswith("srfrd_encoder_") else []
            procs.append((src, obj, subprocess.Popen([hipcc] + g.FLAGS + extra + [f"-DSRFRD_SKEW={m}", "-c", os.path.join(g.CSRC, src), "-o", obj])))
        for src, obj, pr in procs:
            if pr.wait() != 0:
                raise SystemExit(f"hipcc failed on {src} ({m})")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [o for _, o, _ in procs], check=True)
        print("built", lib, flush=True)


def run(extra):
    rc = 0
    masks = os.environ.get("SRFRD_SKEW_MASKS", ",".join(MASKS)).split(",")      # (a subset: SRFRD_SKEW_MASKS=0x0001,0xfffe)
    for m in masks:
        lib = os.path.join(OUT, f"libsrfrd_hip_{m}.so")
        if not os.path.exists(lib):
            raise SystemExit(f"{lib} is missing: run `python tools/race_skew.py build` first")
        env = dict(os.environ, SRFRD_LIB_PATH=lib)
        print(f"=== skew mask {m}", flush=True)
        r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-q", "-p", "no:cacheprovider"] + extra,
                           cwd=ROOT, env=env)
        print(f"=== skew mask {m}: rc {r.returncode}", flush=True)
        rc = rc or r.returncode
    return rc


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    elif len(sys.argv) > 1 and sys.argv[1] == "run":
        sys.exit(run(sys.argv[2:]))
    else:
        raise SystemExit(__doc__)
