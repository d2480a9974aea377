"""Device-resident ray / target generation for `train_rnb`.

The reference prepares every step on the host: `Dataset.ps_gen_random_rays_at_view_on_all_lights`
(models/dataset.py:351-376) draws pixels with CPU `randint`, fancy-indexes the CPU-resident
`[n_images, n_lights, H, W, 3]` image stacks, copies the results to the GPU; `exp_runner.py:214-220` moves the
pixel indices back to the CPU to gather the per-pixel light directions; `near_far_from_sphere`
(dataset.py:448-458) follows.  `DeviceRays` keeps the stacks in HBM (226 MB each for DiLiGenT-MV, 7.5 GB each for
a 200-view 1024^2 capture: 288 GB of HBM hold them all) and produces everything a step consumes with one launch
(`rnb_gen_rays_at_view`).  Same method names, argument meaning and return tuple as the reference's `Dataset`, so
`exp_runner.py:174-180` works unchanged on it; `pixels_x` / `pixels_y` may be passed in so that tests can use the
reference's own draws."""
from __future__ import annotations

import ctypes as C

import torch

from . import native


class DeviceRays:
    def __init__(self, images, images_warmup, masks, light_directions, light_directions_warmup, intrinsics_all_inv,
                 pose_all, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("DeviceRays: the image stacks must live on the GPU (there is no CPU path)")

        def put(t):
            return None if t is None else t.to(device=dev, dtype=torch.float32).contiguous()

        self.images = put(images)                                   # [V, L, H, W, 3]
        self.images_warmup = put(images_warmup)                     # [V, L, H, W, 3] or None
        self.masks = put(masks)                                     # [V, H, W, Cm]
        self.light_directions = put(light_directions)               # [V, L, H, W, 3] or None
        self.light_directions_warmup = put(light_directions_warmup)  # [V, L, 3] or None
        self.intrinsics_all_inv = put(intrinsics_all_inv)           # [V, 4, 4]
        self.pose_all = put(pose_all)                               # [V, 4, 4]
        self.device = dev
        self.n_images, self.n_lights, self.H, self.W = self.images.shape[:4]
        if self.masks.dim() == 3:
            self.masks = self.masks.unsqueeze(-1)
        if self.masks.shape[:3] != (self.n_images, self.H, self.W):
            raise ValueError(f"masks {tuple(self.masks.shape)} do not match images {tuple(self.images.shape)}")

    @classmethod
    def from_dataset(cls, dataset, device="cuda"):
        """Takes the tensors of a constructed reference `Dataset` (models/dataset.py:219-239)."""
        return cls(dataset.images, getattr(dataset, "images_warmup", None), dataset.masks,
                   getattr(dataset, "light_directions", None), getattr(dataset, "light_directions_warmup", None),
                   dataset.intrinsics_all_inv, dataset.pose_all, device)

    # ------------------------------------------------------------------------------------------------
    def _pixels(self, batch_size, pixels_x, pixels_y):
        if pixels_x is None:
            # dataset.py:356-357 (drawn on the device here: the host generator is not on the path)
            pixels_x = torch.randint(0, self.W, (batch_size,), device=self.device)
            pixels_y = torch.randint(0, self.H, (batch_size,), device=self.device)
        else:
            if pixels_x.shape != (batch_size,) or pixels_y.shape != (batch_size,):
                raise ValueError("pixels_x / pixels_y must hold batch_size indices")
            on_host = not pixels_x.is_cuda and not pixels_y.is_cuda
            if on_host:   # host indices are checked on the host (an IndexError, like torch indexing)
                if batch_size and (int(pixels_x.min()) < 0 or int(pixels_x.max()) >= self.W
                                   or int(pixels_y.min()) < 0 or int(pixels_y.max()) >= self.H):
                    raise IndexError(f"pixel index out of range for a {self.H} x {self.W} image")
            pixels_x = pixels_x.to(device=self.device, dtype=torch.int64).contiguous()
            pixels_y = pixels_y.to(device=self.device, dtype=torch.int64).contiguous()
            if not on_host:   # device indices: asynchronous device-side check, no host round trip
                ok = (pixels_x >= 0) & (pixels_x < self.W) & (pixels_y >= 0) & (pixels_y < self.H)
                torch._assert_async(ok.all())
        return pixels_x, pixels_y

    def _launch(self, img_idx, pixels_x, pixels_y, want_rgb, want_warmup, want_lights, want_near_far):
        B = pixels_x.numel()
        L = self.n_lights
        v = int(img_idx)
        if not 0 <= v < self.n_images:
            raise IndexError(f"img_idx {v} out of range (n_images {self.n_images})")
        f32 = dict(dtype=torch.float32, device=self.device)
        data = torch.empty(B, 7, **f32)
        rgb = torch.empty(L, B, 3, **f32) if want_rgb else None
        rgb_wu = torch.empty(L, B, 3, **f32) if want_warmup else None
        lights = torch.empty(L, B, 3, **f32) if want_lights else None
        near = torch.empty(B, 1, **f32) if want_near_far else None
        far = torch.empty(B, 1, **f32) if want_near_far else None
        if want_warmup and self.images_warmup is None:
            raise ValueError("DeviceRays was built without images_warmup")
        if want_lights and self.light_directions is None:
            raise ValueError("DeviceRays was built without light_directions")
        with native.on_device(data) as stream:
            native.check(native.load().rnb_gen_rays_at_view(
                native.ptr(self.intrinsics_all_inv[v]), native.ptr(self.pose_all[v]),
                native.ptr(self.images[v]) if want_rgb else None,
                native.ptr(self.images_warmup[v]) if want_warmup else None,
                native.ptr(self.masks[v]), self.masks.shape[-1],
                native.ptr(self.light_directions[v]) if want_lights else None,
                native.ptr(pixels_x), native.ptr(pixels_y), B, L, self.H, self.W, native.ptr(data), native.ptr(rgb),
                native.ptr(rgb_wu), native.ptr(lights), native.ptr(near), native.ptr(far), stream))
        return data, rgb, rgb_wu, lights, near, far

    # ------------------------------------------------------------------ the reference's Dataset methods
    def ps_gen_random_rays_at_view_on_all_lights(self, img_idx, batch_size, pixels_x=None, pixels_y=None):
        """models/dataset.py:351-376: returns (data [B,7], images_warmup [L,B,3], images [L,B,3], pixels_x,
        pixels_y), all on the device."""
        px, py = self._pixels(batch_size, pixels_x, pixels_y)
        data, rgb, rgb_wu, _, _, _ = self._launch(img_idx, px, py, True, self.images_warmup is not None, False, False)
        return data, rgb_wu, rgb, px, py

    def near_far_from_sphere(self, rays_o, rays_d):
        """models/dataset.py:448-458 (torch ops on device tensors, as in the reference)."""
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return mid - 1.0, mid + 1.0

    def light_directions_at(self, img_idx, pixels_y, pixels_x):
        """exp_runner.py:218: `light_directions[cbn, :, pixels_y, pixels_x, :]` -> [L, B, 3], without the host
        round trip of the pixel indices."""
        px, py = self._pixels(pixels_x.numel(), pixels_x, pixels_y)
        return self._launch(img_idx, px, py, False, False, True, False)[3]

    # ------------------------------------------------------------------ everything for one step, one launch
    def sample(self, img_idx, batch_size, warmup=False, pixels_x=None, pixels_y=None):
        """Inputs of one `train_rnb` step (exp_runner.py:174-220) as a dict: rays_o, rays_d, near, far, mask,
        true_rgb, lights_dir (shaped for `render_rnb` / `render_rnb_warmup`), pixels_x, pixels_y."""
        px, py = self._pixels(batch_size, pixels_x, pixels_y)
        data, rgb, rgb_wu, lights, near, far = self._launch(img_idx, px, py, not warmup, warmup, not warmup, True)
        if warmup:
            if self.light_directions_warmup is None:
                raise ValueError("DeviceRays was built without light_directions_warmup")
            lights_dir = self.light_directions_warmup[int(img_idx)].reshape(self.n_lights, 1, 1, 3)
        else:
            lights_dir = lights.reshape(self.n_lights, batch_size, 1, 3)
        return {"rays_o": data[:, :3], "rays_d": data[:, 3:6], "mask": data[:, 6:7], "near": near, "far": far,
                "true_rgb": rgb_wu if warmup else rgb, "lights_dir": lights_dir, "pixels_x": px, "pixels_y": py}
