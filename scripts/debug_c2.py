import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import __graft_entry__ as ge
ge.load_package()
from conftest import Golden, rel_err, sample
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.utils.detfill import det_input, fill_module_, ce_cotangent
ce = len(sys.argv) > 2 and sys.argv[2] == 'ce'
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
G = Golden("swin_unetr_c2"); tag = "c2_m0"; c = G.meta["cases"][tag]
n = lambda k: parse_normalization(k, True, 4, 2)
m = SwinUNETR((96,96,96),1,6,feature_size=48,num_heads=(3,6,12,24),vit_norm_name=n("instance_cond"),encoder_norm_name=n("instance_cond"),decoder_norm_name=n("instance"))
fill_module_(m); m = m.cuda().set_compute_dtype(dtype)
x = det_input(1234, c["x"]).cuda()
y = m(x, c["modalities"])
print("logits", rel_err(sample(y), G.t(f"{tag}/logits_samples")))
y.backward(ce_cotangent(y) if ce else det_input(4321, tuple(y.shape)).cuda())
named = dict(m.named_parameters())
errs = []
for k, g in (G.grads2(tag) if ce else G.grads(tag)).items():
    got = named[k].grad
    errs.append((rel_err(sample(got), g), k, float(g.norm()), g.numel()))
errs.sort(reverse=True)
import statistics
print("median err", statistics.median(e[0] for e in errs))
for e in errs[:40]: print("%.3e %-62s |g| %.3e n %d" % e)
big = [e for e in errs if e[3] >= 4096]
print("big params: median", statistics.median(e[0] for e in big), "max", big[0])
print("---- selected")
for e in errs:
    if e[3] >= 2000 and ("blocks.0" in e[1] or "conv1.conv" in e[1] or "reduction" in e[1] or "transp" in e[1]): print("%.3e %-62s |g| %.3e n %d" % e)
print("---- smallest"); 
for e in errs[-8:]: print("%.3e %-62s |g| %.3e n %d" % e)
