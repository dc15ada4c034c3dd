#!/usr/bin/env python3
"""Ablation timing of the two ragged encoder kernels: which parts of a launch carry its time.

Each variant is the shipped kernel with ONE part compiled out (its results are wrong; only its time is read).  The time a
part's removal takes off the launch is what that part costs on the critical path of the launch - parts overlap, so the
numbers do not add up to the launch.

    python tools/ablate.py --build          (here: patched COPIES of the kernel sources under gpurun_out/ablate/, one
                                             library per variant in srfrd_amd/lib/libabl_<kernel>_<part>.so)
    python tools/ablate.py --run            (GPU box: bench.py per library through SRFRD_LIB_PATH, one table per kernel)
    python tools/ablate.py --clean          (removes the libraries)

The tree's sources are not modified: the patched .inc and a copy of the .hip that includes it live in a scratch directory,
everything else comes from srfrd_amd/csrc through -I.  The patches are textual (anchored on the statements they disable)
and fail loudly when an anchor no longer matches.
"""
import argparse
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "srfrd_amd", "csrc")
LIBD = os.path.join(ROOT, "srfrd_amd", "lib")
WORK = os.path.join(ROOT, "gpurun_out", "ablate")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-math-errno", "-freciprocal-math",
         "-fassociative-math", "-fno-signed-zeros", "-fno-trapping-math", "-fapprox-func", "-ffp-contract=fast"]


def sub1(s, a, b, count=1):
    assert s.count(a) == count, f"anchor matches {s.count(a)}x (expected {count}): {a[:70]!r}"
    return s.replace(a, b)


def resub(s, pat, repl, at_least=1):
    s2, n = re.subn(pat, repl, s)
    assert n >= at_least, f"pattern matched {n}x: {pat!r}"
    return s2


# ---- backward: part -> patch
def bwd_dw_gemms(s):          # the seven weight-gradient products of a block (MFMA chains AND their slab stores)
    return sub1(s, "      f32x4 acc[2];\n#pragma unroll\n      for (int j = 0; j < 2; ++j) acc[j] = f32x4{pre.v",
                "      return;\n      f32x4 acc[2];\n#pragma unroll\n      for (int j = 0; j < 2; ++j) acc[j] = f32x4{pre.v")


def bwd_dw_stores(s):         # only the slab stores (the products stay: an opaque test keeps them alive)
    return sub1(s, "      if (sc.ok) {\n#pragma unroll\n        for (int j = 0; j < 2; ++j)",
                "      if (sc.ok && acc[0][0] + acc[1][1] == 1234.5f) {\n#pragma unroll\n        for (int j = 0; j < 2; ++j)")


def bwd_token_gemms(s):       # the per-token products against packed weights (dA1, dh2, do, dLN1, dx)
    s = resub(s, r"\n(\s+)(rag_packed2?\(mt0,)", r"\n\1if (false) \2", 5)
    return s


def bwd_attention(s):         # dPd, dv, dS, dq, dk
    s = sub1(s, "        const int u = wave + j * nw;\n        if (u < ntri) {", "        const int u = wave + j * nw;\n        if (false) {", 2)
    s = sub1(s, "        const int kt = mt0 + g_w + 2 * j;\n        if (kt < MT) {\n          f32x4 acc1",
             "        const int kt = mt0 + g_w + 2 * j;\n        if (false) {\n          f32x4 acc1", 2)
    return sub1(s, "        const int mt = mt0 + g_w + 2 * j;\n        if (mt < MT) {", "        const int mt = mt0 + g_w + 2 * j;\n        if (false) {")


def bwd_ln_bwd(s):
    return resub(s, r"\n(\s+)(ln_bwd_rows<)", r"\n\1if (false) \2", 3)


def bwd_ln_recompute(s):
    return resub(s, r"\n(\s+)(ln_rows\(nw, X)", r"\n\1if (false) \2", 2)


def bwd_col_sums(s):
    return sub1(s, "      if (wave < NT) {\n        f32x4 acc[1]", "      if (false) {\n        f32x4 acc[1]")


def bwd_checkpoint_loads(s):
    return resub(s, r"\n(\s+)(g_\w+\.load(?:2d)?\()", r"\n\1if (false) \2", 8)


def bwd_logit_gathers(s):
    return sub1(s, "      if (a.table16 != nullptr) chunks(std::true_type{});\n      else chunks(std::false_type{});", "")


def bwd_table_atomics(s):
    s = sub1(s, "          if (pid != 0 && dp != 0.f) atomicAdd", "          if (false) atomicAdd")
    s = sub1(s, "          if (nid != 0 && dn != 0.f) atomicAdd", "          if (false) atomicAdd")
    s = sub1(s, "          else if (id != 0) atomicAdd(&a.grad_table", "          else if (false) atomicAdd(&a.grad_table")
    return sub1(s, "          atomicAdd(&slab[ly.off_pos + t * di + c], gv);", "")


def bwd_dropout(s):
    s = sub1(s, "A2[r * DS + c] = g * drop_mul(ds2, r, c);", "A2[r * DS + c] = g;")
    return sub1(s, "if (is_sas) gv *= drop_mul(dsE, t, c);", "")


BWD = {"dw_gemms": bwd_dw_gemms, "dw_stores_only": bwd_dw_stores, "token_gemms": bwd_token_gemms, "attention": bwd_attention,
       "ln_backward": bwd_ln_bwd, "ln_recompute": bwd_ln_recompute, "col_sums": bwd_col_sums,
       "checkpoint_loads": bwd_checkpoint_loads, "logit_gathers": bwd_logit_gathers, "table_atomics": bwd_table_atomics,
       "dropout_hashes": bwd_dropout}


# ---- forward
def fwd_checkpoint_stores(s):
    s = sub1(s, "      auto gstore4 = [&](float* plane, int rb, int c, const float (&v)[4]) {\n",
             "      auto gstore4 = [&](float* plane, int rb, int c, const float (&v)[4]) {\n        return;\n")
    s = sub1(s, "        if (do_save) a.save_x[x_off(0, b, ly.n_blocks, L, D) + t * D + c] = v;", "")
    return sub1(s, "            if (sv_o && (unsigned)(r - SH) < (unsigned)L)", "            if (false)")


def fwd_dropout(s):
    return s.replace("drop_site(drop_on,", "drop_site(false,").replace("drop_site(drop_on && is_sas,", "drop_site(false,")


def fwd_token_gemms(s):       # q, k + v, out-projection, both FFN products (with their epilogues and checkpoint stores)
    return resub(s, r"\n(\s+)(rag_packed(?:_dual)?\(m)", r"\n\1if (false) \2", 5)


def fwd_attention(s):         # q k^T and P v
    s = sub1(s, "          for (int u = wave; u < ntri; u += nw) {", "          for (int u = wave; false; u += nw) {")
    s = sub1(s, "          if (ma >= 0) gemm_group<2, 1>", "          if (false) gemm_group<2, 1>")
    return sub1(s, "          if (mb >= 0) gemm_group<2, 1>", "          if (false) gemm_group<2, 1>")


def fwd_softmax(s):
    return sub1(s, "        rag_softmax(bXS,", "        if (false) rag_softmax(bXS,")


def fwd_ln(s):
    return resub(s, r"\n(\s+)(ln_rows\(nw)", r"\n\1if (false) \2", 3)


def fwd_head(s):              # hidden-state stores, pos / neg row gathers, logits, loss sums
    return sub1(s, "    if (lo_h) {\n      for (int c = tid; c < dout; c += nthr) a.hidden",
                "    if (true) {} else if (lo_h) {\n      for (int c = tid; c < dout; c += nthr) a.hidden")


def fwd_embedding(s):
    return sub1(s, "      for (; t < L;) {\n        const int id = s_in[t + SH];", "      for (; false;) {\n        const int id = s_in[t + SH];")


FWD = {"checkpoint_stores": fwd_checkpoint_stores, "dropout_hashes": fwd_dropout, "token_gemms": fwd_token_gemms,
       "attention": fwd_attention, "softmax": fwd_softmax, "layernorms": fwd_ln, "head": fwd_head, "embedding": fwd_embedding}

KERNELS = {"bwd": ("srfrd_encoder_bwd_ragged", BWD), "fwd": ("srfrd_encoder_fwd_ragged", FWD)}


def build():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = sorted(glob.glob(os.path.join(LIBD, "obj", "*.o")))
    assert objs, "run __graft_entry__.build() first"
    procs = []
    for kname, (src, parts) in KERNELS.items():
        text = open(os.path.join(CSRC, src + "_kernel.inc")).read()
        def everything(s, parts=parts):
            for n, f in parts.items():
                if n != "dw_stores_only":
                    s = f(s)
            return s
        variants = {"none": lambda s: s, **parts, "all": everything}
        for part, patch in variants.items():
            d = os.path.join(WORK, f"{kname}_{part}")
            os.makedirs(d, exist_ok=True)
            open(os.path.join(d, src + "_kernel.inc"), "w").write(patch(text))
            # (the backward's .hip also includes the forward's .inc for kRagSH / rag_take: the tree's copy, through -I)
            open(os.path.join(d, src + ".hip"), "w").write(open(os.path.join(CSRC, src + ".hip")).read())
            obj = os.path.join(d, src + ".o")
            lib = os.path.join(LIBD, f"libabl_{kname}_{part}.so")
            rest = [o for o in objs if os.path.basename(o) != src + ".o"]
            cmd = (f"{hipcc} {' '.join(FLAGS)} -I{CSRC} -c {os.path.join(d, src + '.hip')} -o {obj} 2> {d}/build.log && "
                   f"{hipcc} --offload-arch=gfx950 -shared -fPIC -o {lib} {' '.join(rest)} {obj}")
            procs.append((f"{kname}:{part}", subprocess.Popen(cmd, shell=True)))
            if len(procs) >= (os.cpu_count() or 4):
                label, p = procs.pop(0)
                assert p.wait() == 0, f"build failed: {label}"
    for label, p in procs:
        assert p.wait() == 0, f"build failed: {label}"
    print("built", len(glob.glob(os.path.join(LIBD, "libabl_*.so"))), "libraries")


def run(steps):
    for kname, (src, parts) in KERNELS.items():
        key = "srfrd_encoder_" + kname
        rows = []
        for part in ["none"] + list(parts) + ["all", "none"]:
            lib = os.path.join(LIBD, f"libabl_{kname}_{part}.so")
            env = dict(os.environ, SRFRD_LIB_PATH=lib)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-metric-parity",
                                  "--no-secondary", "--steps", str(steps), "--warmup", "30"], env=env, capture_output=True, text=True,
                                 timeout=300)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            rows.append((part, d["roofline"]["kernel_ms"][key] * 1e3, d["ms_per_step"] * 1e3))
        base = 0.5 * (rows[0][1] + rows[-1][1])
        print(f"== {key}: {base:.1f} us per launch with nothing removed (HIP events inside bench.py, {steps} steps; first / last row)")
        print(f"   {'part compiled out':22s} {'launch us':>10s} {'saved us':>9s} {'step us':>9s}")
        for part, k, st in rows:
            print(f"   {part:22s} {k:10.1f} {base - k:9.1f} {st:9.1f}")
        sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--clean", action="store_true")
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    if a.build:
        build()
    if a.run:
        run(a.steps)
    if a.clean:
        for f in glob.glob(os.path.join(LIBD, "libabl_*.so")):
            os.remove(f)


if __name__ == "__main__":
    main()
