"""How much of the step is under-fill of the small levels?  One UNet forward over b rows blocks (b = 2: the CFG pair of one prompt, the
bench's step; b = 4: two prompts' pairs in one plan) at the bench shape, same weights, generic forward (no fused CFG / DDIM update:
elementwise, < 0.1 ms).  If forward(4) < 2 x forward(2) by a margin, a two-prompt batch is a serving-side throughput lever."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from videomv_amd.registry import MODEL
import videomv_amd  # noqa: F401
from videomv_amd.camera import entrance_camera_data

H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "40x64").split("x"))
dev = torch.device("cuda", 0)
with torch.device(dev):
    m = MODEL.build(dict(type="UNetSD_T2VBase", y_dim=1024, use_camera_condition=True, use_lgm_refine=False, **bench.FULL))
bench.randomize_(m, 1234)
m.eval()
g = torch.Generator(device=dev).manual_seed(11)
cam1 = entrance_camera_data(24, elevation=15, camera_distance=2.0).to(dev)


def run(b, reps=6, warm=2):
    x = torch.randn(b, 4, 24, H, W, generator=g, device=dev)
    y = torch.randn(b, 77, 1024, generator=g, device=dev)
    cam = cam1.reshape(1, -1, cam1.shape[-1]).repeat(b, 1, 1)
    t = torch.full((b,), 501, dtype=torch.long, device=dev)
    with torch.no_grad():
        for _ in range(warm):
            out = m(x, t, y=y, camera_data=cam)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = m(x, t, y=y, camera_data=cam)
        torch.cuda.synchronize()
    ms = 1000 * (time.perf_counter() - t0) / reps
    return ms, bool(torch.isfinite(out).all())


for order in (2, 4, 2, 4):
    ms, fin = run(order)
    print(f"latent 24x{H}x{W}  b={order}: {ms:.2f} ms per forward = {ms / order:.2f} ms per row block  finite={fin}", flush=True)
