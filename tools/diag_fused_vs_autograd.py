import copy, torch, srfrd_amd
torch.manual_seed(0)
m = srfrd_amd.SASRec(50_000, 50, 50, 0.5, 2, 1, "cuda")
for _, p in m.named_parameters():
    if p.dim() >= 2: torch.nn.init.xavier_normal_(p.data)
m = m.cuda()
u, seq, rsq, pos, prs, neg, nrs = srfrd_amd.synthetic_batch(50_000, 50, 512, seed=1, device="cuda")
m1, m2 = copy.deepcopy(m), copy.deepcopy(m)
m1.dropout_rate = m2.dropout_rate = 0.0
m1.train(); m2.train()
tr = srfrd_amd.FusedTrainer(m1, 512, 50, use_graph=True)
opt = torch.optim.Adam(m2.parameters(), lr=1e-3, betas=(0.9, 0.98))
crit = torch.nn.BCEWithLogitsLoss()
for step in range(3):
    l1 = tr.step(u, seq, rsq, pos, prs, neg, nrs)
    h, pl, nl = m2(u, seq, rsq, pos, prs, neg, nrs)
    idx = torch.where(pos != 0)
    l2 = crit(pl[idx], torch.ones_like(pl)[idx]) + crit(nl[idx], torch.zeros_like(nl)[idx])
    opt.zero_grad(); l2.backward(); opt.step()
    print(step, float(l1), float(l2))
sd1, sd2 = m1.state_dict(), m2.state_dict()
for k in sd1:
    d = (sd1[k] - sd2[k]).abs()
    print(f"{k:45s} max {float(d.max()):.2e} mean {float(d.mean()):.2e} frac>1e-4 {float((d>1e-4).float().mean()):.2e}")
