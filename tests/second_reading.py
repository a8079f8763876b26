"""A SECOND, independent restatement of the mesh path's geometry — written directly from the reference's source text in plain Python floats
(IEEE binary64, math.sqrt; no numpy arithmetic, no FMA), without looking at oracle/oracle.cpp — so that tests/test_second_reading.py can hold
the C++ oracle against it bit for bit.  The reference has no tests and its dragon mesh is absent, so nothing it holds pins
`AccGrid::intersects`; two readings of the same source that agree on every ray are the next best thing.

    AABB::intersects                      core/src/geometry/primitives/aabb.rs:10-31
    Triangle::intersects                  core/src/geometry/primitives/triangle.rs:11-44
    Triangle::get_surface_properties      core/src/geometry/primitives/triangle.rs:47-68
    AccGrid::intersects                   core/src/geometry/acc_grid.rs:89-185

cgmath 0.17 as the reference uses it: dot = (x x' + y y') + z z' (mul_element_wise().sum()), magnitude = sqrt(dot(v, v)),
distance(a, b) = magnitude(b - a), normalize(v) = v * (1 / magnitude), cross the usual determinant form, `1.0 / v` and div_element_wise
component by component; `cast::<i32>()` (num-traits) truncates toward zero and is None — the reference then panics on unwrap() — for NaN or
values outside i32; f64::signum is +1 for +0.0 and -1 for -0.0 and NaN for NaN; f64::min / max return the other operand when one is NaN;
`as usize` of a negative i32 sign-extends and the index arithmetic wraps in a release build.
Test infrastructure: imported by tests only.
"""
import ctypes
import ctypes.util
import math

MASK64 = (1 << 64) - 1


class Panic(Exception):
    """Where the reference would panic (unwrap() of a failed cast)."""


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def scale(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def cross(a, b):
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def sqrt(x):
    return math.sqrt(x) if x >= 0.0 else float("nan")  # IEEE: sqrt of a negative is NaN (math.sqrt raises); sqrt(NaN) is NaN


def magnitude(a):
    return sqrt(dot(a, a))


def distance(a, b):
    return magnitude(sub(b, a))


def div(a, b):
    """IEEE division, including by zero."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return float("nan")
        return math.copysign(float("inf"), a) * math.copysign(1.0, b)


def normalize(a):
    return scale(a, div(1.0, magnitude(a)))


def fmin(a, b):  # f64::min
    if a != a:
        return b
    if b != b:
        return a
    return a if a < b else b


def fmax(a, b):  # f64::max
    if a != a:
        return b
    if b != b:
        return a
    return a if a > b else b


def cast_i32(v):
    if v != v or not (-2147483649.0 < v < 2147483648.0):
        raise Panic("cast::<i32>() of %r" % (v,))
    return int(v)  # truncates toward zero


def signum(v):
    if v != v:
        return v
    return math.copysign(1.0, v)


# ---------------------------------------------------------------- aabb.rs:10-31
def aabb_intersects(bmin, bmax, origin, direction):
    inv = tuple(div(1.0, d) for d in direction)
    t1 = (bmin[0] - origin[0]) * inv[0]
    t2 = (bmax[0] - origin[0]) * inv[0]
    tmin, tmax = fmin(t1, t2), fmax(t1, t2)
    for i in (1, 2):
        t1 = (bmin[i] - origin[i]) * inv[i]
        t2 = (bmax[i] - origin[i]) * inv[i]
        tmin = fmax(tmin, fmin(t1, t2))
        tmax = fmin(tmax, fmax(t1, t2))
    if not (tmax > fmax(tmin, 0.0)):
        return None
    return tmin


# ---------------------------------------------------------------- triangle.rs:11-44
def triangle_intersects(v0, v1, v2, origin, direction):
    eps = 0.00000001
    edge1, edge2 = sub(v1, v0), sub(v2, v0)
    h = cross(direction, edge2)
    a = dot(edge1, h)
    if a < eps and a > -eps:
        return None
    f = div(1.0, a)
    s = sub(origin, v0)
    u = f * dot(s, h)
    if u < 0.0 or u > 1.0:
        return None
    q = cross(s, edge1)
    v = f * dot(direction, q)
    if v < 0.0 or u + v > 1.0:
        return None
    t = f * dot(edge2, q)
    if t > eps:
        return t
    return None


# ---------------------------------------------------------------- triangle.rs:47-68
def triangle_normal(p, n, origin, direction, dist):
    """p = (p0, p1, p2) positions, n = (n0, n1, n2) vertex normals."""

    def area(a, b, c):
        ab, ac, bc = distance(a, b), distance(a, c), distance(b, c)
        s = div(ab + ac + bc, 2.0)
        return sqrt(s * (s - ab) * (s - ac) * (s - bc))

    position = add(origin, scale(direction, dist))
    abc = area(p[0], p[1], p[2])
    abp = area(p[0], p[1], position)
    bcp = area(p[0], p[2], position)
    ba, bb = div(abp, abc), div(bcp, abc)
    bc = 1.0 - (ba + bb)
    normal = add(add(scale(n[2], ba), scale(n[1], bb)), scale(n[0], bc))
    return normalize(normal)


# ---------------------------------------------------------------- acc_grid.rs:89-185
def grid_intersects(grid, origin, direction):
    """grid: dict with bbox_min, bbox_max, cell_size (3-tuples), resolution (3 ints), cells (offsets into mapping_table), mapping_table,
    tri_pos (list of 9-tuples).  Returns (distance, triangle index) or None; None as well where the reference would panic."""
    try:
        bmin = grid["bbox_min"]
        outer = aabb_intersects(bmin, grid["bbox_max"], origin, direction)
        if outer is None:
            return None
        outer_pos = add(origin, scale(direction, outer))
        cs = grid["cell_size"]
        start = sub(origin, bmin)
        cell = [cast_i32(div(start[i], cs[i])) for i in range(3)]
        if cell[0] < 0 or cell[1] < 0 or cell[2] < 0:
            start = sub(outer_pos, bmin)
            cell = [cast_i32(div(start[i], cs[i])) for i in range(3)]
        step = [cast_i32(signum(d)) for d in direction]
        t_delta = [div((-cs[i] if direction[i] < 0.0 else cs[i]), direction[i]) for i in range(3)]
        t_max = [div((float(cell[i] + (0 if direction[i] < 0.0 else 1)) * cs[i]) - start[i], direction[i]) for i in range(3)]
    except Panic:
        return None
    res = grid["resolution"]
    cells, table, tris = grid["cells"], grid["mapping_table"], grid["tri_pos"]
    while True:
        x, y, z = cell[0] & MASK64, cell[1] & MASK64, cell[2] & MASK64  # `as usize`
        index = (x + res[0] * ((y + z * res[2]) & MASK64)) & MASK64     # Q5: res.z where res.y is meant; wrapping arithmetic
        if index >= len(cells):
            return None
        c = int(cells[index])
        count = int(table[c])
        closest, closest_hit = 5712515.0, None
        for i in range(1, count + 1):
            ti = int(table[c + i])
            p = tris[ti]
            d = triangle_intersects(p[0:3], p[3:6], p[6:9], origin, direction)
            if d is not None and d < closest:
                closest, closest_hit = d, (d, ti)
        if closest_hit is not None:
            return closest_hit
        if t_max[0] < t_max[1]:
            axis = 0 if t_max[0] < t_max[2] else 2
        else:
            axis = 1 if t_max[1] < t_max[2] else 2
        cell[axis] += step[axis]
        if cell[axis] >= res[axis] or cell[axis] < 0:
            return None
        t_max[axis] += t_delta[axis]


# ================================================================ the rest of the path: Scene::intersect, the primitives, trace(), ray generation
#     Sphere::intersects / get_surface_properties   core/src/geometry/primitives/sphere.rs:11-35
#     Plane::intersects / get_surface_properties    core/src/geometry/primitives/plane.rs:11-32
#     Scene::intersect                              core/src/scene.rs:54-74
#     trace, generate_primary_ray[_with_dof], BRDF helpers, samplers, ONB        src/trace.rs:232-416
# Random numbers: the reference draws from rand::random::<f64>() (unseedable); this repo defines the stream (include/raymond_hip.h, "RNG"):
# Philox4x32-10 (Random123), key = seed, counter = (pixel, sample, block, 0); block 0 is the pixel jitter, every round of the lens rejection loop
# takes one block, every shaded depth one block (r1, r2 = its two 53-bit uniforms) and `r` (:260) is the 22-bit uniform of the sample's
# previous block.  Philox is restated here from the Random123 paper, not from the repo's C++.
# libm: Python's math.acos / tan / pow are the C library's, as are the oracle's std:: calls.  NdotH.powf(2.0) is x * x (LLVM folds
# pow(x, 2.0) unconditionally); (1 - cos).powf(5.0) is a libm call.  `theta.sin()` and `theta.cos()` of ONE angle (:291, :403): on a
# linux-gnu target LLVM merges the pair into one sincos() call (SelectionDAG FSINCOS when the libcall exists) and gcc does the same to the
# oracle; glibc 2.35's sincos() is not always sin() and cos() — about 0.1 % of the arguments of this path differ by an ulp in one of the two —
# so this reading calls sincos() itself.  (On a target without sincos the reference's samples would differ from these in the last digits:
# its results were never pinned to a libm.)
F_MAX = 1.7976931348623157e308  # core/src/math.rs:20
PI = 3.14159265358979323846     # core/src/math.rs:19


_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
_libm.sincos.restype = None


def sincos(x):
    s, c = ctypes.c_double(), ctypes.c_double()
    _libm.sincos(x, ctypes.byref(s), ctypes.byref(c))
    return s.value, c.value


def philox4x32_10(counter, key):
    c0, c1, c2, c3 = counter
    k0, k1 = key
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


class Stream:
    """The random numbers of one (pixel, sample)."""

    def __init__(self, seed, pixel, sample):
        self.key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
        self.pixel, self.sample, self.block, self.u22 = pixel, sample, 0, 0.0

    def next_block(self):
        w0, w1, w2, w3 = philox4x32_10((self.pixel, self.sample, self.block, 0), self.key)
        self.block += 1
        first = float(((w1 << 32) | w0) >> 11) * (1.0 / 9007199254740992.0)
        second = float(((w3 << 32) | w2) >> 11) * (1.0 / 9007199254740992.0)
        previous_u22 = self.u22
        self.u22 = float(((w0 & 0x7FF) << 11) | (w2 & 0x7FF)) * (1.0 / 4194304.0)
        return first, second, previous_u22


def neg(a):
    return (-a[0], -a[1], -a[2])


def hadamard(a, b):
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


def vdiv(a, s):
    return (div(a[0], s), div(a[1], s), div(a[2], s))


def sphere_intersects(center, radius, origin, direction):
    c = sub(center, origin)
    t = dot(c, direction)
    q = sub(c, scale(direction, t))  # c - t * ray.direction
    p = dot(q, q)
    if p > radius * radius:
        return None
    t -= sqrt(radius * radius - p)
    if t <= 0.0:
        return None
    return t


def plane_intersects(p_origin, normal, origin, direction):
    denom = dot(normal, neg(direction))
    if denom > 1e-6:
        p0l0 = sub(p_origin, origin)
        t = div(dot(p0l0, neg(normal)), denom)
        if t >= 0.0:
            return t
    return None


def scene_intersect(objects, origin, direction):
    """objects: list of dicts {kind: 'plane'|'sphere'|'grid', ..., material: (kind, color, roughness)} -> (index, distance, triangle) or None"""
    closest, found = F_MAX, None
    for i, o in enumerate(objects):
        sub_index = 0
        if o["kind"] == "plane":
            d = plane_intersects(o["origin"], o["normal"], origin, direction)
        elif o["kind"] == "sphere":
            d = sphere_intersects(o["origin"], o["radius"], origin, direction)
        else:
            r = grid_intersects(o["grid"], origin, direction)
            d, sub_index = (None, 0) if r is None else r
        if d is not None and d < closest:
            closest, found = d, (i, d, sub_index)
    return found


def lerp(mn, mx, a):
    return mn + a * (mx - mn)


def create_coordinate_system_of_n(n):
    sign = 1.0 if n[2] > 0.0 else -1.0
    a = div(-1.0, sign + n[2])
    b = n[0] * n[1] * a
    return (1.0 + sign * n[0] * n[0] * a, sign * b, -sign * n[0]), (b, sign + n[1] * n[1] * a, -n[1])


def mat3_from_cols_mul(c0, c1, c2, v):  # cgmath Matrix3::from_cols(c0, c1, c2) * v = c0 * v.x + c1 * v.y + c2 * v.z
    return add(add(scale(c0, v[0]), scale(c1, v[1])), scale(c2, v[2]))


def ggx_distribution(n, h, roughness):
    a2 = roughness * roughness
    ndh = dot(n, h)
    den = (ndh * ndh) * (a2 - 1.0) + 1.0
    den = fmax(PI * den * den, 1e-7)
    return div(a2, den)


def geometry_schlick_ggx(n, v, r):
    num = fmax(dot(n, v), 0.0)
    k = div(r * r, 8.0)
    return div(num, num * (1.0 - k) + k)


def geometry_smith(n, v, l, r):
    return geometry_schlick_ggx(n, v, r) * geometry_schlick_ggx(n, l, r)


def powf(x, y):
    try:
        return math.pow(x, y)
    except (OverflowError, ValueError):
        return float("nan") if x != x or x < 0.0 else float("inf")


def fresnel_schlick(cos_theta, f0):
    p = powf(1.0 - cos_theta, 5.0)
    return add(f0, scale(sub((1.0, 1.0, 1.0), f0), p))


def libm1(f, x):
    try:
        return f(x)
    except ValueError:
        return float("nan")


def trace(scene, settings, stream, origin, direction, depth, path=None):
    """src/trace.rs:232-320.  scene: {objects, camera_position}; settings: {bounce_limit}; path: optional list that receives (object, triangle)."""
    if depth > settings["bounce_limit"]:
        return (0.0, 0.0, 0.0)
    found = scene_intersect(scene["objects"], origin, direction)
    if found is None:
        if path is not None:
            path.append((-1, 0))
        return (0.0, 0.0, 0.0)
    oi, dist, tri = found
    if path is not None:
        path.append((oi, tri))
    o = scene["objects"][oi]
    if o["kind"] == "plane":
        normal = o["normal"]
    elif o["kind"] == "sphere":
        normal = normalize(sub(add(origin, scale(direction, dist)), o["origin"]))
    else:
        g = o["grid"]
        p, n = g["tri_pos"][tri], g["tri_nrm"][tri]
        normal = triangle_normal((p[0:3], p[3:6], p[6:9]), (n[0:3], n[3:6], n[6:9]), origin, direction, dist)
    fragment_position = add(origin, scale(direction, dist))
    kind, color, roughness = o["material"]
    if kind == "emission":
        return color
    metalness = 0.0 if kind == "diffuse" else 1.0
    view_dir = normalize(sub(scene["camera_position"], fragment_position))
    f0 = (lerp(0.04, color[0], metalness), lerp(0.04, color[1], metalness), lerp(0.04, color[2], metalness))
    r1, r2, r = stream.next_block()  # r: of the previous block; r1, r2: this depth's
    tangent, bitangent = create_coordinate_system_of_n(normal)
    prob_d = lerp(0.5, 0.0, metalness)
    if r < prob_d:
        theta = libm1(math.acos, sqrt(r1))
        phi = 2.0 * PI * r2
        pdf = sqrt(r1)
        (st, ct), (sp, cp) = sincos(theta), sincos(phi)
        sample = (st * cp, ct, st * sp)
        sample_world = normalize(mat3_from_cols_mul(tangent, normal, bitangent, sample))
        radiance = trace(scene, settings, stream, add(fragment_position, scale(normal, 0.00001)), sample_world, depth + 1, path)
        cos_theta = fmax(dot(normal, sample_world), 0.0)
        halfway = normalize(add(sample_world, view_dir))
        fresnel = fresnel_schlick(fmax(dot(halfway, view_dir), 0.0), f0)
        diffuse_part = scale(sub((1.0, 1.0, 1.0), fresnel), 1.0 - metalness)
        output = scale(hadamard(hadamard(diffuse_part, color), radiance), cos_theta)
        return vdiv(output, prob_d * pdf)
    reflect = normalize(sub(neg(view_dir), scale(scale(normal, -dot(view_dir, normal)), 2.0)))  # -v - 2.0 * (-(v . n) * n)
    a = roughness * roughness
    phi = 2.0 * PI * r1
    theta = a * sqrt(div(r2, 1.0 - r2))
    (st, ct), (sp, cp) = sincos(theta), sincos(phi)
    h = (st * cp, ct, st * sp)
    t2, b2 = create_coordinate_system_of_n(reflect)
    sample_world = normalize(mat3_from_cols_mul(t2, reflect, b2, h))
    radiance = trace(scene, settings, stream, add(fragment_position, scale(normal, 0.0001)), sample_world, depth + 1, path)
    cos_theta = dot(normal, sample_world)
    light_dir = normalize(sample_world)
    halfway = normalize(add(light_dir, view_dir))
    F = fresnel_schlick(dot(halfway, view_dir), f0)
    D = ggx_distribution(normal, halfway, roughness)
    G = geometry_smith(normal, view_dir, sample_world, roughness)
    nominator = scale(F, D * G)  # D * G * F
    denominator = 4.0 * dot(normal, view_dir) * cos_theta + 0.001
    specular = vdiv(nominator, denominator)
    output = scale(hadamard(specular, radiance), cos_theta)
    pdf = div(D * dot(normal, halfway), 4.0 * dot(halfway, view_dir)) + 0.0001
    return vdiv(vdiv(output, 1.0 - prob_d), pdf)


def generate_primary_ray(x, y, cam, stream):
    width, height = float(cam["width"]), float(cam["height"])
    aspect = div(width, height)
    u0, u1, _ = stream.next_block()
    fx = float(x) + (u0 - 0.5)
    fy = float(y) + (u1 - 0.5)
    tan_half = math.tan(cam["fov_vert"] / 2.0 * PI / 180.0)
    px = (2.0 * div(fx + 0.5, width) - 1.0) * tan_half * aspect
    py = (1.0 - 2.0 * div(fy + 0.5, height)) * tan_half
    return cam["position"], normalize((px, py, 1.0))


def generate_primary_ray_with_dof(x, y, cam, stream):
    p_origin, p_dir = generate_primary_ray(x, y, cam, stream)
    pos = cam["position"]
    while True:
        a, b, _ = stream.next_block()
        r1, r2 = a * 2.0 - 1.0, b * 2.0 - 1.0
        start = (pos[0] + r1 * cam["aperture_radius"], pos[1] + r2 * cam["aperture_radius"], pos[2])
        if distance(start, pos) < cam["aperture_radius"]:
            break
    focal_origin = add(pos, scale((0.0, 0.0, 1.0), cam["focal_length"]))
    d = plane_intersects(focal_origin, (0.0, 0.0, -1.0), p_origin, p_dir)
    if d is None:
        raise Panic("unwrap() of the focal-plane hit")
    end = add(pos, scale(p_dir, d))  # position + distance * direction
    return start, normalize(sub(end, start))


def sample_pixel(scene, cam, settings, seed, x, y, sample, use_dof=False, path=None):
    """One execution of the worker's inner loop body (src/trace.rs:199-200; :199 is the pinhole ray, the thin lens is this repo's opt-in)."""
    stream = Stream(seed, y * cam["width"] + x, sample)
    try:
        origin, direction = (generate_primary_ray_with_dof if use_dof else generate_primary_ray)(x, y, cam, stream)
    except Panic:
        return (0.0, 0.0, 0.0)
    return trace(scene, settings, stream, origin, direction, 1, path)


# ================================================================ the output stage (cli_old/src/main.rs:161-181, after TaskHandle::await's division, src/trace.rs:95)
def resolve_pixel(accum_rgb, sample_count, exposure=1.0, gamma=2.2):
    """-> (r, g, b) as the 8-bit values the reference writes: 1 - exp(-p * exposure), powf(1 / gamma), * 255, Vector3::cast::<u8>() — None (the
    pixel keeps its initial (0, 0, 0)) when ANY component is NaN or outside (-1, 256); the cast truncates toward zero."""
    out = []
    for a in accum_rgb:
        p = div(a, sample_count)
        try:
            e = math.exp(p * -1.0 * exposure)
        except OverflowError:
            e = float("inf")
        tm = 1.0 - e
        tm = powf(tm, div(1.0, gamma))
        out.append(tm * 255.0)
    if any(v != v or not (-1.0 < v < 256.0) for v in out):
        return (0, 0, 0)
    return tuple(int(v) for v in out)
