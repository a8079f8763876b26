"""Host-side mirror of the reference's render API over the C-ABI.

`render_tiled(scene, settings) -> TaskHandle` keeps the reference's shape (src/trace.rs:137-230,
TaskHandle :70-135): the framebuffer is split into tiles in the reference's column-major order
(:142-173), tiles are handed to workers, finished tiles come back as `Message.TileFinished` and
`TaskHandle.await_()` assembles the W*H image divided by the sample count (:82-113).  The one
difference is what a worker is: a GPU context that runs rmd_render_tiles over its share of the
tiles (the per-pixel body :197-205 for ALL pixels of those tiles at once) instead of an OS thread
looping over pixels.
"""
import ctypes as C

import numpy as np

from . import abi
from . import lib as _lib
from .scene import generate_tiles, tile_array


class Context:
    """rmd_context: one per GPU."""

    def __init__(self, device=0, stream=None):
        self.L = _lib.load()
        self.handle = C.c_void_p()
        if stream is None:
            _lib.check(self.L.rmd_context_create(device, C.byref(self.handle)))
        else:
            _lib.check(self.L.rmd_context_create_on_stream(device, C.c_void_p(stream), C.byref(self.handle)))
        self.device = device

    def close(self):
        if self.handle:
            self.L.rmd_context_destroy(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def check(self, status):
        _lib.check(status, self.handle)

    def set_tunable(self, key, value):
        """rmd_context_set_tunable: scheduling knobs (abi.RMD_TUNE_*); no setting changes a result.  0 = the library's choice."""
        self.check(self.L.rmd_context_set_tunable(self.handle, key, int(value)))

    def get_tunable(self, key):
        v = C.c_int64()
        self.check(self.L.rmd_context_get_tunable(self.handle, key, C.byref(v)))
        return v.value

    def memory_info(self):
        """(free, total) bytes of the device's memory."""
        free, total = C.c_uint64(), C.c_uint64()
        self.check(self.L.rmd_context_memory_info(self.handle, C.byref(free), C.byref(total)))
        return free.value, total.value

    def synchronize(self):
        self.check(self.L.rmd_context_synchronize(self.handle))

    def last_launch_info(self):
        """rmd_last_launch_info: how the most recent render was launched (passes, split_k, persistent, end_black_paths, has_grid)."""
        info = abi.LaunchInfo()
        self.check(self.L.rmd_last_launch_info(self.handle, C.byref(info)))
        return info

    def last_kernel_ms(self):
        ms = C.c_float()
        self.check(self.L.rmd_last_kernel_ms(self.handle, C.byref(ms)))
        return ms.value


class DeviceScene:
    """rmd_scene: the Scene resident in one GPU's HBM."""

    def __init__(self, ctx, scene):
        self.ctx = ctx
        objs, n, descs, ng, keep = scene.flatten()
        self.handle = C.c_void_p()
        ctx.check(ctx.L.rmd_scene_create(ctx.handle, objs, n, descs, ng, C.byref(self.handle)))

    def close(self):
        if self.handle:
            self.ctx.L.rmd_scene_destroy(self.handle)
            self.handle = C.c_void_p()


class Framebuffer:
    """W*H*3 f64 accumulation buffer in HBM (library-allocated, or wrapping a caller's device pointer)."""

    def __init__(self, ctx, width, height, device_ptr=None):
        self.ctx, self.width, self.height = ctx, width, height
        self.n = width * height * 3
        self.owned = device_ptr is None
        if self.owned:
            p = C.c_void_p()
            ctx.check(ctx.L.rmd_framebuffer_alloc(ctx.handle, width, height, C.byref(p)))
            self.ptr = p
        else:
            self.ptr = C.c_void_p(device_ptr)

    def zero(self):
        self.ctx.check(self.ctx.L.rmd_framebuffer_zero(self.ctx.handle, self.ptr, self.n))

    def download(self):
        out = np.empty((self.height, self.width, 3), dtype=np.float64)
        self.ctx.check(self.ctx.L.rmd_framebuffer_download(self.ctx.handle, self.ptr, out.ctypes.data_as(C.c_void_p), self.n))
        return out

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert arr.size == self.n
        self.ctx.check(self.ctx.L.rmd_framebuffer_upload(self.ctx.handle, arr.ctypes.data_as(C.c_void_p), self.ptr, self.n))

    def download_tiles(self, tiles):
        """rmd_framebuffer_download_tiles: the pixels of `tiles` (left, top, width, height) as ONE packed array, tile after tile, each
        row-major (height, width, 3) — core::tile::Tile.data's layout; returns the list of per-tile views."""
        arr = tile_array(tiles)
        n = sum(w * h for (_, _, w, h) in tiles)
        out = np.empty(n * 3, dtype=np.float64)
        self.ctx.check(self.ctx.L.rmd_framebuffer_download_tiles(self.ctx.handle, self.ptr, self.width, self.height, arr, len(tiles), out.ctypes.data_as(C.c_void_p)))
        views, at = [], 0
        for (_, _, w, h) in tiles:
            views.append(out[at : at + w * h * 3].reshape(h, w, 3))
            at += w * h * 3
        return views

    def upload_tiles(self, tiles, datas):
        """rmd_framebuffer_upload_tiles: the inverse (datas: one (height, width, 3) array per tile)."""
        packed = np.ascontiguousarray(np.concatenate([np.asarray(d, dtype=np.float64).reshape(-1) for d in datas]))
        self.ctx.check(self.ctx.L.rmd_framebuffer_upload_tiles(self.ctx.handle, packed.ctypes.data_as(C.c_void_p), self.ptr, self.width, self.height, tile_array(tiles), len(tiles)))

    def close(self):
        if self.owned and self.ptr:
            self.ctx.L.rmd_framebuffer_free(self.ctx.handle, self.ptr)
            self.ptr = C.c_void_p()


def render_tiles(ctx, dscene, camera_settings, settings, tiles, framebuffer, sample_begin=0, sample_count=None, sync=True):
    """rmd_render_tiles[_async]: add `sample_count` samples per pixel of `tiles` into `framebuffer`."""
    cam = camera_settings.pod()
    st = settings.pod(sample_begin, sample_count)
    arr = tiles if isinstance(tiles, tuple) and len(tiles) == 2 and hasattr(tiles[0], "_length_") else (tile_array(tiles), len(tiles))
    fn = ctx.L.rmd_render_tiles if sync else ctx.L.rmd_render_tiles_async
    ctx.check(fn(ctx.handle, dscene.handle, C.byref(cam), C.byref(st), arr[0], arr[1], framebuffer.ptr))


def resolve_tonemap(ctx, framebuffer, sample_count, exposure=1.0, gamma=2.2):
    """TaskHandle::await's divide + cli_old's tone-map/gamma/u8 cast (cli_old/src/main.rs:161-181) -> (H, W, 3) uint8."""
    out = np.empty((framebuffer.height, framebuffer.width, 3), dtype=np.uint8)
    ctx.check(
        ctx.L.rmd_resolve_tonemap(ctx.handle, framebuffer.ptr, framebuffer.width, framebuffer.height, sample_count, exposure, gamma,
                                  out.ctypes.data_as(C.c_void_p))
    )
    return out


# ---------------------------------------------------------------- the reference-shaped API
class Tile:  # core/src/tile.rs:7-14
    def __init__(self, left, top, width, height, sample_count, data):
        self.left, self.top, self.width, self.height = left, top, width, height
        self.sample_count = sample_count
        self.data = data  # (height, width, 3) running sums, like Tile.data


class Message:  # src/trace.rs:62-66
    def __init__(self, kind, tile):
        self.kind, self.tile = kind, tile

    @staticmethod
    def TileFinished(tile):
        return Message("TileFinished", tile)

    @staticmethod
    def TileProgressed(tile):
        return Message("TileProgressed", tile)


class TaskHandle:  # src/trace.rs:70-135
    def __init__(self, settings, messages):
        self.settings = settings
        self._messages = list(messages)
        self.callback = None

    def set_callback(self, callback):
        self.callback = callback

    def poll(self):
        return self._messages.pop(0) if self._messages else None

    def async_await(self):
        while self._messages and self._messages[0].kind == "TileProgressed":
            m = self._messages.pop(0)
            if self.callback:
                self.callback(m.tile)

    def await_(self):
        """`await`: W*H radiance values, row-major, each the tile sum divided by its sample count (:93-99)."""
        cam = self.settings.camera_settings
        out = np.zeros((cam.backbuffer_height, cam.backbuffer_width, 3), dtype=np.float64)
        while self._messages:
            m = self._messages.pop(0)
            if m.kind != "TileFinished":
                break  # the reference stops collecting at the first non-TileFinished message (:101-103)
            t = m.tile
            out[t.top : t.top + t.height, t.left : t.left + t.width] = t.data / float(t.sample_count)
        return out


def render_tiled(scene, settings, devices=(0,)):
    """render_tiled (src/trace.rs:137): tiles -> workers -> TaskHandle.  Workers are GPU contexts."""
    cam = settings.camera_settings
    W, H = cam.backbuffer_width, cam.backbuffer_height
    tiles = generate_tiles(W, H, settings.tile_size)
    workers = []
    for d in devices:
        ctx = Context(d)
        workers.append((ctx, DeviceScene(ctx, scene), Framebuffer(ctx, W, H)))
    shares = [tiles[i :: len(workers)] for i in range(len(workers))]
    step = settings.samples_per_iteration if settings.samples_per_iteration else settings.sample_count
    messages = []
    done = 0
    try:
        while done < settings.sample_count:
            n = min(step, settings.sample_count - done)
            for (ctx, ds, fb), share in zip(workers, shares):
                if share:
                    render_tiles(ctx, ds, cam, settings, share, fb, done, n, sync=False)
            for ctx, _, _ in workers:
                ctx.synchronize()
            done += n
            if done < settings.sample_count and settings.samples_per_iteration:
                for (ctx, ds, fb), share in zip(workers, shares):
                    img = fb.download()
                    for (l, t, w, h) in share:
                        messages.append(Message.TileProgressed(Tile(l, t, w, h, done, img[t : t + h, l : l + w].copy())))
        finished = []
        for (ctx, ds, fb), share in zip(workers, shares):
            img = fb.download()
            for (l, t, w, h) in share:
                finished.append(Message.TileFinished(Tile(l, t, w, h, settings.sample_count, img[t : t + h, l : l + w].copy())))
        messages = messages + finished  # progress snapshots first, then the finished tiles
    finally:
        for ctx, ds, fb in workers:
            fb.close()
            ds.close()
            ctx.close()
    handle = TaskHandle(settings, messages)
    return handle
