import sys, torch, time
sys.path.insert(0, '.')
import srfrd_amd
torch.manual_seed(0)
I, L, B = 50000, 50, 512
m = srfrd_amd.SASRec(I, L, 50, 0.5, 2, 1, "cuda")
for _, p in m.named_parameters():
    if p.dim() >= 2: torch.nn.init.xavier_normal_(p.data)
m = m.cuda().train()
tr = srfrd_amd.FusedTrainer(m, B, L, slots=8)
# a learnable synthetic task: fixed pool of 8 batches -> loss must go down
for i in range(8):
    tr.ids_ring[i].copy_(srfrd_amd.synthetic_batch(I, L, B, seed=1, index=i, device="cuda", packed=True)[1])
losses = []
t0 = time.time()
for s in range(3000):
    loss = tr.step_slot(s % 8)
    if s % 300 == 0 or s == 2999:
        losses.append(float(loss.cpu()))
print("losses", [round(x, 4) for x in losses], "time", round(time.time() - t0, 2))
assert all(x == x and x < 5 for x in losses), "nan / blow-up"
assert losses[-1] < losses[0] - 0.2, "no learning"
sd = m.state_dict()
assert all(torch.isfinite(v).all() for v in sd.values())
print("soak ok")
