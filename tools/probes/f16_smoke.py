import sys, torch
sys.path.insert(0, "/root/repo")
from l4p_amd.weights import ModelCfg, seeded_state_dict
from tests.golden_utils import make_batch
from tests.test_encoder_dpt_gpu import build
TASKS = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]
cfg = ModelCfg.mini(); sd = seeded_state_dict(cfg)
batch = make_batch(32, 6)
outs = {}
for prec in ("32-true", "bf16", "16-mixed"):
    m = build(cfg, sd, prec)
    with torch.no_grad():
        o = m.forward({k: v.clone() for k, v in batch.items()}, TASKS)
    torch.cuda.synchronize()
    outs[prec] = {k: v.float().cpu() for k, v in o.items() if torch.is_tensor(v)}
for k in outs["32-true"]:
    r = outs["32-true"][k]
    print(k, {p: f"{float((outs[p][k]-r).norm()/(r.norm()+1e-30)):.2e}" for p in ("bf16", "16-mixed")})
