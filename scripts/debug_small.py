import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import __graft_entry__ as ge
ge.load_package()
from conftest import Golden, rel_err, sample
from mi_seg_amd.networks.nets.swin_unetr import SwinUNETR
from mi_seg_amd.networks.norms.utils import parse_normalization
from mi_seg_amd.utils.detfill import det_input, fill_module_
G = Golden("swin_unetr_small"); tag = "fs12_64_m10"; c = G.meta["cases"][tag]
n = lambda k: parse_normalization(k, True, 4, 2)
m = SwinUNETR((64,64,64),1,6,feature_size=12,num_heads=(3,6,12,24),downsample=c["downsample"],vit_norm_name=n(c["vit_norm"]),encoder_norm_name=n(c["encoder_norm"]),decoder_norm_name=n(c["decoder_norm"]))
fill_module_(m); m = m.cuda()
x = det_input(1234, c["x"]).cuda()
y = m(x, c["modalities"])
print("logits", rel_err(sample(y), G.t(f"{tag}/logits_samples")))
y.backward(det_input(4321, tuple(y.shape)).cuda())
named = dict(m.named_parameters())
errs = []
for k, g in G.grads(tag).items():
    got = named[k].grad
    errs.append((rel_err(sample(got), g), k, float(g.norm()), float(sample(got).norm())))
errs.sort(reverse=True)
for e in errs[:25]: print("%.3e %-60s want %.3e got %.3e" % e)
