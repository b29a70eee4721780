"""Which intermediate of the forward's tail differs between a hipGraph replay and the eager run once the input has new values?"""
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from openseg3d_amd import batch as B, config, ops, scene, segformer
dev = torch.device("cuda:0")
cfg = config.default_cfg(); ds = config.DatasetSpec(cfg)
torch.manual_seed(0)
model = segformer.build_segmentor(cfg, ds).to(dev).eval()
pts = B.collate_points([scene.make_scene(3)[::4]], dev)
n = pts.shape[0]
stash = {}
def tap(name, mod):
    orig = mod.forward
    def fwd(*a, **k):
        y = orig(*a, **k)
        stash[name + ".in"] = a[0].clone()
        stash[name + ".out"] = y.clone()
        return y
    mod.forward = fwd
for name in ("point_encoder", "fusion_encoder", "se", "classifier", "vfe"):
    tap(name, getattr(model, name))
tap("se.fc", model.se.fc)
tap("se.fc0", model.se.fc[0])
tap("se.fc2", model.se.fc[2])
orig_gather = ops.gather_rows
def g2(f, i, s=None):
    y = orig_gather(f, i, s); stash["gather.in"] = f.clone(); stash["gather.out"] = y.clone(); return y
ops.gather_rows = g2
with torch.no_grad():
    batch = model.prepare_batch(B.batch_from_resident(pts, [n], ds.voxel_size, ds.point_cloud_range))
    model(dict(batch))
    side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        model(dict(batch))
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(dict(batch))["point_out"]
    gstash = dict(stash)
    g.replay(); torch.cuda.synchronize()
    batch["points"][:, 5:] = torch.rand_like(batch["points"][:, 5:])
    stash.clear()
    want = model(dict(batch))["point_out"].clone()
    estash = dict(stash)
    g.replay(); torch.cuda.synchronize()
    print("point_out equal", torch.equal(out, want), float((out - want).abs().max()))
    for k in estash:
        print(f"{k:22s} graph vs eager max diff {float((gstash[k] - estash[k]).abs().max()):.3e}")
