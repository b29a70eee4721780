"""Where does the device Lovasz gradient differ from the reference module's? (GPU box only)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from openseg3d_amd import losses

d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "losses.npz"))
dev = torch.device("cuda:0")
for name, fn in (("lovasz", losses.LovaszLoss()), ("lovasz_all", losses.LovaszLoss(classes="all")),
                 ("ohem", losses.OHEMCrossEntropyLoss(keep_thresh=0.7))):
    x = torch.from_numpy(d["logits"]).to(dev).requires_grad_(True)
    loss = fn(x, torch.from_numpy(d["labels"]).to(dev))
    loss.backward()
    g, r = x.grad.cpu().numpy(), d[name + "_grad"]
    e = np.abs(g - r)
    bad = np.argwhere(e > 1e-9 + 1e-5 * np.abs(r).max())
    print(name, "loss", float(loss), float(d[name]), "max err", e.max(), "max grad", np.abs(r).max(), "bad entries", len(bad),
          "rows", sorted(set(bad[:, 0].tolist()))[:20], "rel L2", np.linalg.norm(g - r) / np.linalg.norm(r))
    p = torch.softmax(torch.from_numpy(d["logits"]), 1).numpy()
    for row in sorted(set(bad[:, 0].tolist()))[:6]:
        lab = d["labels"][row]
        print("  row", row, "label", lab, "p[label]", p[row, lab] if lab != 255 else None, "max p", p[row].max(), "cols", bad[bad[:, 0] == row, 1].tolist()[:8])
