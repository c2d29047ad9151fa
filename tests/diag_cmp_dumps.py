import torch, sys
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
mods = {}
for k in a:
    m = k.split(".")[0]
    mods.setdefault(m, []).append(k)
for m, ks in mods.items():
    scale = max(float(a[k].abs().max()) for k in ks)
    w = max((float((a[k] - b[k]).abs().max()) / scale, k) for k in ks)
    print(f"{m:13s} {w[0]:.2e} {w[1]}")
