import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime, rerotation_table, rope_table
cfg = preset("tiny128"); w = make_weights(cfg, dtype=torch.bfloat16, jitter=True)
rt = Runtime(cfg, w, max_step_tokens=256, max_vit_frames=4, max_positions=4096)
W, S, T = 48, 4, 7
st = rt.open_stream("default_sink", W, S)
g = torch.Generator().manual_seed(5)
cos, sin = rope_table(4096, cfg.lm.head_dim, cfg.lm.rope_theta)
for step in range(9):
    bk = st.export_kv(0).cpu(); L = st.get_seq_length()
    x = (torch.randn(1, T, cfg.lm.hidden_size, generator=g) * 0.5).bfloat16()
    rt.lm_step([st], x.cuda())
    if L + T < W: continue
    keep = W - S - T
    rc, rs = rerotation_table(cos, sin, W, S, T)
    ak = st.export_kv(0).cpu()
    kk = bk[:, -keep:]; h = kk.shape[-1] // 2
    rot = torch.cat((-kk[..., h:], kk[..., :h]), dim=-1)
    want = (kk * rc[None]) + (rot * rs[None])
    got = ak[:, S:S + keep]
    bad = (got != want)
    print("step", step, "L", L, "mismatch", bad.sum().item(), "of", bad.numel(), "maxdiff", (got.float() - want.float()).abs().max().item())
    idx = bad.nonzero()[:6]
    for i in idx:
        i = tuple(i.tolist()); d = i[2]
        k1, k2 = kk[i].item(), kk[i[0], i[1], (d + h) % (2 * h)].item()
        print("  ", i, "got", got[i].item(), "want", want[i].item(), "k", k1, "k_pair", k2, "cos", rc[i[1], d].item(), "sin", rs[i[1], d].item())
    print("  unrotated equal to before?", torch.equal(got, kk), " bad by key:", bad.any(-1).any(0).nonzero().flatten().tolist()[:40])
