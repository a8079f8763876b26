"""Scratch timing of a scene whose mesh is small in the frame (few rays enter the grid's box): python tools/sparse_time.py [scale] [spp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymond_amd import render, scenes
from raymond_amd.scene import AccGrid, Grid, Material, Object, Scene, Sphere, generate_tiles

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 50
mesh = scenes.lumpy_sphere_mesh(91, extent=(2.3 * scale, 1.7 * scale, 1.0 * scale), centre=(0.0, 0.15 * scale, 0.0))
mesh.bake_transform((0.0, -0.3, 2.9))
sc = Scene()
sc.objects.append(Object(Sphere((-1.0, -0.5, 3.5), 0.5), Material.Diffuse((1.0, 0.0, 0.0), 0.02)))
sc.objects.append(Object(Grid(AccGrid.build_from_mesh(mesh)), Material.Metal((1.0, 1.0, 0.1), 0.15)))
sc.objects.extend(scenes._room_planes())
st = scenes.config_settings("C3", spp=spp)
cam = st.camera_settings
tiles = generate_tiles(cam.backbuffer_width, cam.backbuffer_height, st.tile_size)
with render.Context(0) as ctx:
    ds = render.DeviceScene(ctx, sc)
    fb = render.Framebuffer(ctx, cam.backbuffer_width, cam.backbuffer_height)
    for it in range(3):
        fb.zero()
        render.render_tiles(ctx, ds, cam, st, tiles, fb)
        ms = ctx.last_kernel_ms()
        print("mesh scale %.2f spp=%d kernel %.1f ms -> %.1f Msamples/s" % (scale, spp, ms, cam.backbuffer_width * cam.backbuffer_height * spp / ms / 1e3), flush=True)
