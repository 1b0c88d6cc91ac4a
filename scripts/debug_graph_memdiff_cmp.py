import pickle, numpy as np
La, da, pa = pickle.load(open("/tmp/dump_good.pkl", "rb")); Lb, db, pb = pickle.load(open("/tmp/dump_bad.pkl", "rb"))
print("same alloc sequence:", [l[2:] for l in La] == [l[2:] for l in Lb])
n = len(Lb)
for i, ((seq, ptr, nb, dt, where), a, b) in enumerate(zip(Lb, da, db)):
    clobbered = any(Lb[j][1] < ptr + nb and ptr < Lb[j][1] + Lb[j][2] for j in range(i + 1, n))
    if clobbered:
        continue
    nd = int((a != b).sum())
    print("seq %4d +%d %s %s  %s" % (seq, nb, dt, ("DIFF bytes %d" % nd) if nd else "same", where))
