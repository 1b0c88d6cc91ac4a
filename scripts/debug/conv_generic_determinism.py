"""bitwise run-to-run check of single kernels on the small bf16 shapes of the 32^3 UNETR / UNet (generic conv path, fused small norms, thin stem):
every repetition must equal the first (debug aid for the open item of DESIGN.md section 3)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
ge.load_package()
from mi_seg_amd.hip import ops, lib as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dt = torch.bfloat16
bad = 0
for S in (4, 8, 16, 32):
    for Cin, Cout in ((8, 8), (16, 8), (16, 16), (32, 16), (32, 32), (64, 32), (48, 48), (96, 48), (8, 16), (16, 32)):
        x = torch.randn(2, S, S, S, Cin, device="cuda").to(dt)
        w = torch.randn(Cout, Cin, 3, 3, 3, device="cuda") / (27 * Cin) ** 0.5
        fwdp, bwdp = ops.pack_conv3(w, dt)
        ref = ops.conv3_fwd(x, fwdp, Cout).clone()
        junk = []
        for r in range(reps):
            if r % 7 == 0:      # disturb the allocator / caches a little
                junk = [torch.randn(1 + (r * 977) % 100000, device="cuda") for _ in range(3)]
            y = ops.conv3_fwd(x, fwdp, Cout)
            if not torch.equal(y, ref):
                bad += 1
                d = (y.float() - ref.float()).abs()
                print(f"conv {S}^3 {Cin}->{Cout} rep {r}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}", flush=True)
        # instance norm (fused small / chunked) on the conv output
        B, Sv = 2, S ** 3
        styles = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
        gam = [torch.rand(Cout, device="cuda") + 0.5 for _ in range(2)]
        bet = [torch.rand(Cout, device="cuda") for _ in range(2)]
        ops.begin_step()
        yn, _ = ops.instnorm_fwd(ref, B, Sv, styles, gam, bet, act=L.ACT_LEAKY)
        yn = yn.clone()
        for r in range(reps):
            ops.begin_step()
            y2, _ = ops.instnorm_fwd(ref, B, Sv, styles, gam, bet, act=L.ACT_LEAKY)
            if not torch.equal(y2, yn):
                bad += 1
                d = (y2.float() - yn.float()).abs()
                print(f"norm {S}^3 C {Cout} rep {r}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}", flush=True)
print("done, mismatches:", bad)
