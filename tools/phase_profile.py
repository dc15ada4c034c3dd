#!/usr/bin/env python3
"""Per-phase cycle breakdown of the fused encoder kernels (diagnostic; run on the GPU box).

Builds a copy of srfrd_encoder.hip with a STAMP(n) after every workgroup barrier of the two kernels
(-DSRFRD_STAMPS), runs the C2 workload once through forward and backward, and prints for each barrier-delimited
phase the mean s_memtime ticks per workgroup and its share.  Never used by the product, tests or bench.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "srfrd_amd", "csrc")
OUT = os.path.join(ROOT, "srfrd_amd", "lib", "libsrfrd_hip_stamps.so")


def build():
    labels = {}
    tmp = "/tmp/srfrd_stamps"
    os.makedirs(tmp + "/srfrd_amd/csrc", exist_ok=True)
    os.makedirs(tmp + "/include", exist_ok=True)
    for f in os.listdir(CSRC):
        open(f"{tmp}/srfrd_amd/csrc/{f}", "w").write(open(os.path.join(CSRC, f)).read())
    for which in ("fwd", "bwd", "bwd_slots", "bwd_chunks", "fwd_ragged", "bwd_ragged"):
        fname = f"srfrd_encoder_{which}_kernel.inc"
        out = _stamp_file(os.path.join(CSRC, fname), which, labels)
        open(f"{tmp}/srfrd_amd/csrc/{fname}", "w").write("\n".join(out))
    open(f"{tmp}/include/srfrd_hip.h", "w").write(open(os.path.join(ROOT, "include", "srfrd_hip.h")).read())
    srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    import __graft_entry__ as ge
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-munsafe-fp-atomics"] + ge.ENCODER_FLAGS + [
           "-DSRFRD_STAMPS", "-o", OUT] + [f"{tmp}/srfrd_amd/csrc/{f}" for f in srcs]
    subprocess.run(cmd, check=True)
    import json
    json.dump({f"{k[0]}:{k[1]}": v for k, v in labels.items()}, open(OUT + ".labels.json", "w"), indent=0)


def _stamp_file(path, which, labels):
    src = open(path).read()
    lines = src.split("\n")
    out, n, kernel = [], 0, None
    for ln, line in enumerate(lines):
        if line == "}" and kernel:
            out.append("  STAMP_FLUSH;")
        out.append(line)
        if "encoder_%s_kernel(const EncArgs a) {" % which in line:
            kernel, n = which, 0
        elif line == "}":
            kernel = None
        if kernel and line.strip() == "__syncthreads();" and n < 120:
            # only inside the sequence loop (after STAMP_INIT of this kernel)
            seen_init = any("STAMP_INIT" in l for l in out[-4000:] if True) and _after_init(out, kernel)
            if seen_init:
                prev = next((l.strip() for l in reversed(lines[:ln]) if l.strip() and l.strip() != "__syncthreads();"), "")
                out.append(f"    STAMP({n});")
                labels[(kernel, n)] = f"L{ln + 1}: {prev[:90]}"
                n += 1
    return out


def _after_init(out, kernel):
    # walk back to the kernel header; True if a STAMP_INIT line was emitted after it
    for l in reversed(out):
        if l.strip() == "STAMP_INIT":
            return True
        if "_kernel(const EncArgs a) {" in l:
            return False
    return False


def run():
    import json
    import torch
    os.environ["SRFRD_LIB_PATH"] = OUT
    import srfrd_amd
    from srfrd_amd import _lib
    labels = json.load(open(OUT + ".labels.json"))
    I, L, B, D = 50_000, 50, 512, 50
    torch.manual_seed(0)
    m = srfrd_amd.SASRec(I, L, D, 0.5, 2, 1, "cuda")
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
    m = m.cuda().train()
    _, seq, rsq, pos, prs, neg, nrs = srfrd_amd.synthetic_batch(I, L, B, seed=1, device="cuda")
    ids = m._prep(seq, None, pos, None, neg, None)
    ragged = os.environ.get("SRFRD_NO_RAGGED") is None
    for which in ("fwd", "bwd"):
        dbg = torch.zeros(1024, 128, device="cuda", dtype=torch.int64)
        dbg2 = torch.zeros(1024, 128, device="cuda", dtype=torch.int64)
        out = m._launch_fwd(*ids, 0.5, 7, save=True, dbg=dbg.view(torch.float32) if which == "fwd" else None)
        if which == "bwd":
            dpl = torch.full_like(out["pos_logits"], 0.1)
            m._launch_bwd(*ids, 0.5, 7, out, None, dpl, dpl, dbg=dbg.view(torch.float32))
        torch.cuda.synchronize()
        # duration of the stamped kernels themselves (events), to calibrate ticks -> time
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            if which == "fwd":
                m._launch_fwd(*ids, 0.5, 7, save=True, dbg=dbg2.view(torch.float32))
            else:
                m._launch_bwd(*ids, 0.5, 7, out, None, dpl, dpl, dbg=dbg2.view(torch.float32))
        e1.record()
        torch.cuda.synchronize()
        print(f"   stamped {which} kernel: {e0.elapsed_time(e1) / 5 * 1000:.1f} us per launch")
        d = dbg.cpu().double()
        used = d[d.sum(1) > 0]
        mean = used.mean(0)
        tot = float(mean.sum())
        print(f"== {which}: {used.shape[0]} workgroups, {tot:.0f} ticks (100 MHz -> {tot / 100:.1f} us) per workgroup")
        # the slowest tenth of the workgroups (the long sequences: a launch lasts as long as they do)
        tot_wg = used.sum(1)
        slow = used[tot_wg >= tot_wg.quantile(0.9)].mean(0)
        fast = used[tot_wg <= tot_wg.quantile(0.3)].mean(0)
        print(f"   slowest 10 %: {float(slow.sum()):.0f} ticks; fastest 30 %: {float(fast.sum()):.0f} ticks")
        lab = which + ("_ragged" if ragged else "")
        for i in range(128):
            if mean[i] > 0:
                print(f"  {i:3d} {float(mean[i]):9.0f} {100 * float(mean[i]) / tot:5.1f}%  slow {float(slow[i]):8.0f} fast {float(fast[i]):8.0f}  {labels.get(f'{lab}:{i}', labels.get(f'{which}:{i}', ''))}")


def run_c4(I=200_000, L=100, which="bwd_slots"):
    """seq_len 100 fused training step (BASELINE configs[3] geometry): the slot-placed backward, phase by phase
    (--c5: seq_len 200, the row-chunked backward)"""
    import ctypes as C
    import json
    import torch
    os.environ["SRFRD_LIB_PATH"] = OUT
    import srfrd_amd
    from srfrd_amd import _lib
    from srfrd_amd._lib import check, ptr
    labels = json.load(open(OUT + ".labels.json"))
    B, D = 512, 50
    torch.manual_seed(0)
    m = srfrd_amd.SASRec(I, L, D, 0.5, 2, 1, "cuda")
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            torch.nn.init.xavier_normal_(p.data)
    m = m.cuda().train()
    tr = srfrd_amd.FusedTrainer(m, B, L, use_graph=False)
    tr.refresh()
    tr.ids.copy_(srfrd_amd.synthetic_batch(I, L, B, seed=1, device="cuda", packed=True)[1])
    tr._enqueue_fwd()
    ids, fk, pfk, nfk, p, seed_dev, seq0 = tr._ids_of(0)
    lay_t, tab = m._table_args()

    def bwd(dbg):
        check(_lib.lib().srfrd_encoder_bwd(C.byref(lay_t), tab, tr._dense_ptr(tr.flat), ptr(tr.packed), ptr(ids[0]), ptr(fk), ptr(ids[2]),
                                           ptr(pfk), ptr(ids[4]), ptr(nfk), B, L, p, 0, seed_dev, seq0, ptr(tr.hidden), ptr(tr.pl),
                                           ptr(tr.nl), ptr(tr.save_x), ptr(tr.save_h1), ptr(tr.save_aux), None, None, None, 1,
                                           ptr(tr.grad), None, ptr(tr.slabs), ptr(tr.scratch), tr.n_scratch,
                                           ptr(dbg.view(torch.float32)), 0, tr._stream()), "srfrd_encoder_bwd")
    dbg = torch.zeros(1024, 128, device="cuda", dtype=torch.int64)
    dbg2 = torch.zeros(1024, 128, device="cuda", dtype=torch.int64)
    bwd(dbg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        bwd(dbg2)
    e1.record()
    torch.cuda.synchronize()
    print(f"   stamped slots kernel: {e0.elapsed_time(e1) / 5 * 1000:.1f} us per launch")
    d = dbg.cpu().double()
    used = d[d.sum(1) > 0]
    mean = used.mean(0)
    tot = float(mean.sum())
    print(f"== {which}: {used.shape[0]} workgroups, {tot:.0f} ticks per workgroup")
    for i in range(128):
        if mean[i] > 0:
            print(f"  {i:3d} {float(mean[i]):9.0f} {100 * float(mean[i]) / tot:5.1f}%  {labels.get(f'{which}:{i}', '')}")


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    if "--run" in sys.argv:
        run()
    if "--c4" in sys.argv:
        run_c4()
    if "--c2s" in sys.argv:       # seq_len 50 (BASELINE configs[1]) on the slot-placed backward, two workgroups per CU
        run_c4(50_000, 50, "bwd_slots")
    if "--c5" in sys.argv:
        run_c4(1_000_000, 200, "bwd_chunks")
