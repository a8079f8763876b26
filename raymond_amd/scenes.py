"""Synthetic inputs for the benchmark configurations (BASELINE.json configs).

* `reflective_spheres()`  — the README's ReflectiveSpheres scene, reconstructed from the commented-out
  sphere in cli_old/src/main.rs:56-58 plus the live objects of :48-55 and :77-127.
* `gold_dragon_standin()` — cli_old/src/main.rs:45-127 verbatim, with the missing assets/meshes/dragon_vrip.ply
  replaced by a deterministic procedural closed mesh of ~100k triangles.
The mesh generator uses only IEEE-exact operations (+ - * / sqrt), so it yields bit-identical
triangles on any host.
"""
import numpy as np

from .scene import AccGrid, CameraSettings, Grid, Material, Mesh, Object, Plane, Scene, Settings, Sphere, Transform

SEED = 0x5EED0001


def _room_planes():
    """The six walls of cli_old/src/main.rs:77-127, in source order."""
    return [
        Object(Plane((0.0, -1.0, 0.0), (0.0, 1.0, 0.0)), Material.Diffuse((0.75, 0.75, 0.75), 0.5)),  # floor
        Object(Plane((0.0, 2.0, 0.0), (0.0, -1.0, 0.0)), Material.Emission((1.5, 1.5, 1.5), (1.0, 1.0, 1.0), 0.27, 0.0)),  # ceiling
        Object(Plane((0.0, 0.0, -2.0), (0.0, 0.0, 1.0)), Material.Diffuse((1.0, 1.0, 1.0), 0.4)),  # front wall
        Object(Plane((0.0, 0.0, 5.0), (0.0, 0.0, -1.0)), Material.Diffuse((0.0, 0.0, 0.0), 0.9)),  # back wall
        Object(Plane((-2.0, 0.0, 0.0), (1.0, 0.0, 0.0)), Material.Diffuse((0.0, 0.0, 0.0), 0.3)),  # left wall
        Object(Plane((2.0, 0.0, 0.0), (-1.0, 0.0, 0.0)), Material.Diffuse((0.0, 0.0, 0.0), 0.3)),  # right wall
    ]


def reflective_spheres():
    scene = Scene()
    scene.objects.append(Object(Sphere((-1.0, -0.5, 3.5), 0.5), Material.Diffuse((1.0, 0.0, 0.0), 0.02)))
    scene.objects.append(Object(Sphere((0.74, -0.25, 3.5), 0.75), Material.Metal((0.05, 0.25, 1.0), 0.01)))
    scene.objects.extend(_room_planes())
    return scene


def lumpy_sphere_mesh(n=91, extent=(2.3, 1.7, 1.0), centre=(0.0, 0.15, 0.0)):
    """Closed, smooth-shaded procedural mesh with 12*n*n triangles (n=91 -> 99,372).

    A cube-sphere (lattice points of the cube surface pushed onto the unit sphere) whose radius is
    modulated by Chebyshev-polynomial lobes, then scaled into `extent` and moved to `centre`.
    The defaults reproduce the footprint of the dragon in the reference's examples/GoldDragon.png
    (592x340, f = 170/tan(27.5 deg) = 326.6 px): it spans ~260 x 195 px at z ~ 2.9, i.e. ~2.3 x 1.7
    world units, stands on the floor plane y = -1 after the (0, -0.3, 2.9) bake of cli_old/src/main.rs:61,
    and keeps the Stanford dragon's 0.21 : 0.15 : 0.09 proportions (depth 1.0).  size.z <= size.y, as the
    reference's `res.z` index quirk (Q5) requires of any mesh it can build a grid for.
    Vertex normals are area-weighted face-normal sums.
    """
    idx = {}
    verts = []

    def vid(i, j, k):
        key = (i, j, k)
        v = idx.get(key)
        if v is None:
            v = len(verts)
            idx[key] = v
            verts.append(key)
        return v

    faces = []
    # (fixed axis, fixed value, u axis, v axis) with (u, v, outward) right-handed
    for axis, val, ua, va in ((0, n, 1, 2), (0, 0, 2, 1), (1, n, 2, 0), (1, 0, 0, 2), (2, n, 0, 1), (2, 0, 1, 0)):
        for a in range(n):
            for b in range(n):
                def p(da, db):
                    c = [0, 0, 0]
                    c[axis] = val
                    c[ua] = a + da
                    c[va] = b + db
                    return vid(*c)

                v00, v10, v11, v01 = p(0, 0), p(1, 0), p(1, 1), p(0, 1)
                faces.append((v00, v10, v11))
                faces.append((v00, v11, v01))
    lat = np.asarray(verts, dtype=np.float64)
    faces = np.asarray(faces, dtype=np.int64)
    c = lat * (2.0 / n) - 1.0
    d = c / np.sqrt((c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]) + c[:, 2] * c[:, 2])[:, None]
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    t3 = lambda t: (4.0 * t * t - 3.0) * t  # cos(3 acos t)
    t2 = lambda t: 2.0 * t * t - 1.0
    radius = 1.0 + 0.22 * t3(x) * t3(y) + 0.15 * t2(z) * t3(y) + 0.10 * t3(z) * t2(x)
    p = d * radius[:, None]
    half = np.max(np.abs(p), axis=0)
    p = p * (np.asarray(extent, dtype=np.float64) * 0.5 / half)[None, :] + np.asarray(centre, dtype=np.float64)[None, :]
    p0, p1, p2 = p[faces[:, 0]], p[faces[:, 1]], p[faces[:, 2]]
    e1, e2 = p1 - p0, p2 - p0
    fn = np.stack(
        [e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2], e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]],
        axis=1,
    )
    vn = np.zeros_like(p)
    for k in range(3):
        np.add.at(vn, faces[:, k], fn)
    vn = vn / np.sqrt((vn[:, 0] * vn[:, 0] + vn[:, 1] * vn[:, 1]) + vn[:, 2] * vn[:, 2])[:, None]
    tri_pos = np.concatenate([p0, p1, p2], axis=1)
    tri_nrm = np.concatenate([vn[faces[:, 0]], vn[faces[:, 1]], vn[faces[:, 2]]], axis=1)
    return Mesh(tri_pos, tri_nrm)


def gold_dragon_standin(n=91, grid_builder=None):
    """cli_old/src/main.rs:45-127 with the procedural mesh in place of dragon_vrip.ply.

    `grid_builder(mesh) -> AccGrid` defaults to the product's host builder.
    """
    mesh = lumpy_sphere_mesh(n)
    mesh.bake_transform((0.0, -0.3, 2.9))  # :61
    grid = (grid_builder or AccGrid.build_from_mesh)(mesh)  # :63
    scene = Scene()
    scene.objects.append(Object(Sphere((-1.0, -0.5, 3.5), 0.5), Material.Diffuse((1.0, 0.0, 0.0), 0.02)))  # :48-55
    scene.objects.append(Object(Grid(grid), Material.Metal((1.0, 1.0, 0.1), 0.15)))  # :72-75
    scene.objects.extend(_room_planes())
    return scene


def mesh_scene(mesh, translate=(0.0, -0.3, 2.9), grid_builder=None):
    """cli_old/src/main.rs:45-127 with `mesh` (e.g. one of the reference's assets/meshes/*.ply, loaded by Mesh.load_ply) in
    the dragon's place: load (:60), bake_transform (:61), build_from_mesh (:63), gold Metal material (:72-75)."""
    mesh = Mesh(mesh.tri_pos.copy(), mesh.tri_nrm.copy())
    mesh.bake_transform(translate)
    grid = (grid_builder or AccGrid.build_from_mesh)(mesh)
    scene = Scene()
    scene.objects.append(Object(Sphere((-1.0, -0.5, 3.5), 0.5), Material.Diffuse((1.0, 0.0, 0.0), 0.02)))
    scene.objects.append(Object(Grid(grid), Material.Metal((1.0, 1.0, 0.1), 0.15)))
    scene.objects.extend(_room_planes())
    return scene


def camera(width, height, aperture_radius=0.0):
    """cli_old/src/main.rs:134-141 at the requested resolution."""
    return CameraSettings(width, height, 55.0, Transform.identity(), focal_length=2.5, aperture_radius=aperture_radius)


# name -> (scene factory name, W, H, spp, bounce_limit, aperture)  — SURVEY.md §8d table
CONFIGS = {
    "C1": ("reflective_spheres", 256, 256, 16, 3, 0.0),
    "C2": ("reflective_spheres", 1920, 1080, 500, 5, 0.0),
    "C3": ("gold_dragon_standin", 1920, 1080, 500, 5, 0.0),
    "C4": ("gold_dragon_standin", 3840, 2160, 2000, 8, 0.0),
    "C5": ("gold_dragon_standin", 1920, 1080, 4000, 5, 0.5),
}


def config_settings(name, spp=None):
    _, w, h, s, b, ap = CONFIGS[name]
    return Settings(camera(w, h, ap), sample_count=s if spp is None else spp, tile_size=(32, 32), bounce_limit=b, seed=SEED,
                    use_dof=ap > 0.0)  # C5 names the thin lens explicitly; the reference's own loop never uses it
