"""``GaussianRenderer`` — the LGM branch's splatting renderer on the gfx950 rasteriser (``csrc/raster.hip``).

Mirrors ``core/gs.py:16-94`` of the reference: ``render(gaussians [B,N,14], cam_view, cam_view_proj, cam_pos, bg_color)``
-> ``{"image": [B,V,3,S,S] (clamped to [0,1]), "alpha": [B,V,1,S,S]}``.  The reference rasterises one (sample, view) at a time;
here ALL B * V views of a call go through ONE batched pass (``vmv_gs_batch_*``: one preprocess launch, one scan, one radix sort with
the view in the key, one blend launch over tiles x views) and the host reads the instance total ONCE per call — 48 host round trips
per LGM-refined step in round 4, one now (``VMV_GS_BATCH=0`` restores the per-view loop; both share their device code and give the
same bits).  The reference delegates to the third-party ``diff_gaussian_rasterization`` extension, which is absent and unpinned; the
kernels follow the published forward algorithm (oracle/gs_ref.py, parity unpinned — DESIGN.md §2).
"""
import ctypes as C
import math

import torch

from . import _lib as L
from .ops import _stream_ptr


class GaussianRenderer:
    def __init__(self, output_size=512, fovy=39.6, znear=0.5, zfar=2.5):
        self.size, self.fovy = int(output_size), float(fovy)
        self.tan_half_fov = math.tan(0.5 * math.radians(self.fovy))
        self.proj_matrix = torch.zeros(4, 4, dtype=torch.float32)          # core/gs.py:23-29
        self.proj_matrix[0, 0] = 1 / self.tan_half_fov
        self.proj_matrix[1, 1] = 1 / self.tan_half_fov
        self.proj_matrix[2, 2] = (zfar + znear) / (zfar - znear)
        self.proj_matrix[3, 2] = -(zfar * znear) / (zfar - znear)
        self.proj_matrix[2, 3] = 1
        self._buf = {}
        self.last_num_rendered, self.last_views = [], 0

    def _buffers(self, N, device):
        key = (N, str(device))
        b = self._buf.get(key)
        if b is None:
            lib = L.load()
            sb, so = C.c_size_t(0), C.c_size_t(0)
            L.check(lib.vmv_gs_workspace_bytes(N, 1, C.byref(sb), C.byref(so)), "gs_workspace_bytes")
            grid = (self.size + 15) // 16
            f = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=device)
            b = dict(depth=f(N), xy=f(N, 2), co=f(N, 4), rect=f(N, 4, dt=torch.int32), touched=f(N, dt=torch.int32),
                     offsets=f(N, dt=torch.int32), scan=torch.zeros(max(int(sb.value), 16), dtype=torch.uint8, device=device),
                     ranges=f(grid * grid, 2, dt=torch.int32), cap=0)
            self._buf[key] = b
        return b

    def _grow(self, b, N, n, device):
        if n <= b["cap"]:
            return
        cap = max(int(n * 1.25), 1 << 16)
        sb, so = C.c_size_t(0), C.c_size_t(0)
        L.check(L.load().vmv_gs_workspace_bytes(N, cap, C.byref(sb), C.byref(so)), "gs_workspace_bytes")
        i64 = lambda: torch.zeros(cap, dtype=torch.int64, device=device)
        i32 = lambda: torch.zeros(cap, dtype=torch.int32, device=device)
        b.update(keys=i64(), keys_s=i64(), vals=i32(), vals_s=i32(), cap=cap,
                 sort=torch.zeros(max(int(so.value), 16), dtype=torch.uint8, device=device))

    def _batch_buffers(self, B, V, N, device):
        key = ("batch", B, V, N, str(device))
        b = self._buf.get(key)
        if b is None:
            lib = L.load()
            VN = B * V * N
            bits = lib.vmv_gs_batch_key_bits(B * V, self.size)
            sb, so = C.c_size_t(0), C.c_size_t(0)
            L.check(lib.vmv_gs_batch_workspace_bytes(VN, 1, bits, C.byref(sb), C.byref(so)), "gs_batch_workspace_bytes")
            grid = (self.size + 15) // 16
            f = lambda *shape, dt=torch.float32: torch.zeros(*shape, dtype=dt, device=device)
            b = dict(depth=f(VN), xy=f(VN, 2), co=f(VN, 4), rect=f(VN, 4, dt=torch.int32), touched=f(VN, dt=torch.int32),
                     offsets=f(VN, dt=torch.int32), scan=torch.zeros(max(int(sb.value), 16), dtype=torch.uint8, device=device),
                     ranges=f(B * V * grid * grid, 2, dt=torch.int32), cap=0, bits=bits,
                     total=torch.zeros(1, dtype=torch.int32).pin_memory() if str(device).startswith("cuda") else torch.zeros(1, dtype=torch.int32))
            self._buf[key] = b
        return b

    def _render_batch(self, gaussians, cam_view, cam_view_proj, bg, images, alphas):
        """All B * V views in one pass (vmv.h VmvGsBatchParams)."""
        device = gaussians.device
        B, V = cam_view.shape[:2]
        N, S = gaussians.shape[1], self.size
        lib = L.load()
        buf = self._batch_buffers(B, V, N, device)
        g = gaussians.float().contiguous()
        views = cam_view.to(device, torch.float32).reshape(B * V, 16).contiguous()
        vps = cam_view_proj.to(device, torch.float32).reshape(B * V, 16).contiguous()
        p = L.GsBatchParams()
        p.gaussians, p.B, p.N, p.V, p.size = g.data_ptr(), B, N, V, S
        p.views, p.view_projs, p.tan_half_fov = views.data_ptr(), vps.data_ptr(), self.tan_half_fov
        p.bg[0], p.bg[1], p.bg[2] = bg
        p.depth, p.xy, p.conic_opacity, p.rect = buf["depth"].data_ptr(), buf["xy"].data_ptr(), buf["co"].data_ptr(), buf["rect"].data_ptr()
        p.tiles_touched, p.offsets = buf["touched"].data_ptr(), buf["offsets"].data_ptr()
        p.scan_temp, p.scan_temp_bytes = buf["scan"].data_ptr(), buf["scan"].numel()
        L.check(lib.vmv_gs_batch_preprocess(C.byref(p), _stream_ptr()), "gs_batch_preprocess")
        # the ONE host round trip of the call: the instance total sizes the sort (pinned destination: a 4-byte copy + stream sync)
        buf["total"].copy_(buf["offsets"][B * V * N - 1:], non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
        n = int(buf["total"][0])
        # The scan and `num_rendered` are int32.  The caller bounds B * V * N * tiles below 2^32, so a total past 2^31 - 1 shows up as a
        # NEGATIVE count (never as a small positive one): refuse it, and refuse totals whose key / value buffers (24 B per instance,
        # x 1.25 head-room) would pass the memory budget — the caller then splits the views.
        if n < 0 or n > self.max_batch_instances:
            return False
        self.last_num_rendered.append(n)
        if n > buf["cap"]:
            cap = max(int(n * 1.25), 1 << 16)
            sb, so = C.c_size_t(0), C.c_size_t(0)
            L.check(lib.vmv_gs_batch_workspace_bytes(B * V * N, cap, buf["bits"], C.byref(sb), C.byref(so)), "gs_batch_workspace_bytes")
            i64 = lambda: torch.zeros(cap, dtype=torch.int64, device=device)
            i32 = lambda: torch.zeros(cap, dtype=torch.int32, device=device)
            buf.update(keys=i64(), keys_s=i64(), vals=i32(), vals_s=i32(), cap=cap,
                       sort=torch.zeros(max(int(so.value), 16), dtype=torch.uint8, device=device))
        p.num_rendered = n
        if n > 0:
            p.keys, p.keys_sorted, p.vals, p.vals_sorted = (buf["keys"].data_ptr(), buf["keys_s"].data_ptr(), buf["vals"].data_ptr(),
                                                            buf["vals_s"].data_ptr())
            p.sort_temp, p.sort_temp_bytes = buf["sort"].data_ptr(), buf["sort"].numel()
        p.ranges = buf["ranges"].data_ptr()
        p.out_color, p.out_alpha = images.data_ptr(), alphas.data_ptr()
        L.check(lib.vmv_gs_batch_render(C.byref(p), _stream_ptr()), "gs_batch_render")
        self._keep.append((g, views, vps))    # (referenced by the enqueued launches)
        return True

    def _render_views(self, gaussians, cam_view, cam_view_proj, bg, images, alphas):
        """The batched pass over as many views at a time as its 32-bit instance count and the memory budget allow: all B * V at once when
        they fit (the VideoMV shapes: 2 x 24 views, 21 M instances), else per sample, else halves of a sample's views; a single view that
        still does not fit goes to the per-view entry points.  Chunks are independent renders: images are those of the one-pass call."""
        B, V = cam_view.shape[:2]
        N, tiles = gaussians.shape[1], ((self.size + 15) // 16) ** 2
        if B * V * N * tiles < (1 << 32) and B * V * N < (1 << 31) and B * V <= 65535:
            if self._render_batch(gaussians, cam_view, cam_view_proj, bg, images, alphas):
                return True
        if B > 1:
            return all(self._render_views(gaussians[b:b + 1], cam_view[b:b + 1], cam_view_proj[b:b + 1], bg, images[b:b + 1], alphas[b:b + 1])
                       for b in range(B))
        if V > 1:
            h = V // 2
            return all(self._render_views(gaussians, cam_view[:, a:z], cam_view_proj[:, a:z], bg, images[:, a:z], alphas[:, a:z])
                       for a, z in ((0, h), (h, V)))
        return False

    @torch.no_grad()
    def render(self, gaussians, cam_view, cam_view_proj, cam_pos=None, bg_color=None, scale_modifier=1):
        if scale_modifier != 1:
            raise NotImplementedError("scale_modifier != 1 is not a VideoMV configuration")
        import os
        device = gaussians.device
        B, V = cam_view.shape[:2]
        N, S = gaussians.shape[1], self.size
        lib = L.load()
        bg = [1.0, 1.0, 1.0] if bg_color is None else [float(v) for v in bg_color.reshape(-1)[:3]]
        images = torch.empty(B, V, 3, S, S, dtype=torch.float32, device=device)
        alphas = torch.empty(B, V, 1, S, S, dtype=torch.float32, device=device)
        self.last_num_rendered, self.last_views, self._keep = [], B * V, []
        self.max_batch_instances = min((1 << 31) - 1, int(os.environ.get("VMV_GS_BATCH_MAX_INSTANCES", str(1 << 28))))   # 2^28 x 30 B = 8 GB
        if os.environ.get("VMV_GS_BATCH", "1") != "0":
            if self._render_views(gaussians, cam_view, cam_view_proj, bg, images, alphas):
                return {"image": images, "alpha": alphas}
        buf = self._buffers(N, device)
        self.last_num_rendered, self.last_views = [], B * V
        for b in range(B):
            g = gaussians[b].float().contiguous()
            for v in range(V):
                view = cam_view[b, v].float().contiguous().to(device)
                vp = cam_view_proj[b, v].float().contiguous().to(device)
                p = L.GsParams()
                p.gaussians, p.N, p.size, p.view, p.view_proj = g.data_ptr(), N, S, view.data_ptr(), vp.data_ptr()
                p.tan_half_fov = self.tan_half_fov
                p.bg[0], p.bg[1], p.bg[2] = bg
                p.depth, p.xy, p.conic_opacity, p.rect = (buf["depth"].data_ptr(), buf["xy"].data_ptr(), buf["co"].data_ptr(),
                                                          buf["rect"].data_ptr())
                p.tiles_touched, p.offsets = buf["touched"].data_ptr(), buf["offsets"].data_ptr()
                p.scan_temp, p.scan_temp_bytes = buf["scan"].data_ptr(), buf["scan"].numel()
                L.check(lib.vmv_gs_preprocess(C.byref(p), _stream_ptr()), "gs_preprocess")
                n = int(buf["offsets"][N - 1].item())                       # the one host round trip per view
                self.last_num_rendered.append(n)
                self._grow(buf, N, n, device)
                p.num_rendered = n
                if n > 0:
                    p.keys, p.keys_sorted, p.vals, p.vals_sorted = (buf["keys"].data_ptr(), buf["keys_s"].data_ptr(),
                                                                    buf["vals"].data_ptr(), buf["vals_s"].data_ptr())
                    p.sort_temp, p.sort_temp_bytes = buf["sort"].data_ptr(), buf["sort"].numel()
                p.ranges = buf["ranges"].data_ptr()
                p.out_color, p.out_alpha = images[b, v].data_ptr(), alphas[b, v].data_ptr()
                L.check(lib.vmv_gs_render(C.byref(p), _stream_ptr()), "gs_render")
        return {"image": images, "alpha": alphas}
