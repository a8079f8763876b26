// raymond.hpp — C++ host mirror of the reference's public API above the hot path, over the C-ABI.
//
// The reference's Rust crates keep their Scene/Camera/Material API and their tile scheduler; only the per-tile
// body (src/trace.rs:197-205) moves to the GPU.  No Rust toolchain exists in this image, so the host side is
// written in C++ with the reference's names, argument meaning and call sequence:
//
//     Scene scene;  scene.objects.push_back(Object{Geometry::Sphere(...), Material::Diffuse(...)});      core/src/scene.rs
//     Mesh mesh = Mesh::load_ply(path);  mesh.bake_transform({0,-0.3,2.9});                               core/src/geometry/mesh.rs:48-121
//     auto grid = AccGrid::build_from_mesh(mesh);                                                         core/src/geometry/acc_grid.rs:36
//     Settings settings{...};  TaskHandle h = render_tiled(scene, settings);  auto image = h.await();     src/trace.rs:137, :82
//
// What differs: a worker is a GPU (an rmd_context) instead of an OS thread, and a worker's unit of work is
// "samples_per_iteration samples for a batch of tiles" in one rmd_render_tiles call instead of one sample of one
// tile.  A tile's running sums stay RESIDENT in its worker's device framebuffer between passes; what crosses the bus is what a
// message carries — a pass's TileProgressed snapshots, the TileFinished tiles — downloaded in Tile.data layout on a copy stream
// while the next pass renders.  Failures throw raymond::Error (the reference panics).
#pragma once
#include <array>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/raymond_hip.h"

namespace raymond {

using Vector3 = std::array<double, 3>; // cgmath::Vector3<f64> (core/src/lib.rs:5)

struct Error : std::runtime_error {
	rmd_status status;
	Error(rmd_status s, const std::string &what) : std::runtime_error(what), status(s) {}
};

// core/src/lib.rs:21-26
struct Material {
	uint32_t kind;
	Vector3 color;
	double roughness;
	std::array<double, 5> aux;
	static Material Diffuse(Vector3 color, double roughness) { return {RMD_MAT_DIFFUSE, color, roughness, {}}; }
	static Material Metal(Vector3 color, double roughness) { return {RMD_MAT_METAL, color, roughness, {}}; }
	static Material Emission(Vector3 e, Vector3 v2 = {1, 1, 1}, double f1 = 0, double f2 = 0) {
		return {RMD_MAT_EMISSION, e, 0.0, {v2[0], v2[1], v2[2], f1, f2}};
	}
};

struct Plane { // core/src/geometry/primitives/plane.rs:5-8
	Vector3 origin, normal;
};
struct Sphere { // core/src/geometry/primitives/sphere.rs:5-8
	Vector3 origin;
	double radius;
};

// core/src/geometry/mesh.rs:10-13 — triangles as 9 doubles of positions + 9 of vertex normals
struct Mesh {
	std::vector<double> tri_pos, tri_nrm;
	size_t triangle_count() const { return tri_pos.size() / 9; }
	// mesh.rs:58-121: ASCII PLY; only `element vertex N` is read from the header, vertex lines are
	// `x y z nx ny nz [s t]`, face lines `3 i j k`; faces with another vertex count are dropped.
	static Mesh load_ply(const std::string &path);
	// mesh.rs:48-56
	void bake_transform(Vector3 translate);
};

// core/src/geometry/acc_grid.rs:27-33 (compact layout of rmd_grid_desc); owns its arrays
class AccGrid {
  public:
	static std::shared_ptr<AccGrid> build_from_mesh(const Mesh &mesh); // acc_grid.rs:36-83 via rmd_grid_build_from_mesh
	~AccGrid();
	const rmd_grid_desc &desc() const { return desc_; }

  private:
	AccGrid() = default;
	rmd_grid_build *build_ = nullptr;
	rmd_grid_desc desc_{};
};

// core/src/scene.rs:9-13
struct Geometry {
	uint32_t kind;
	Plane plane{};
	Sphere sphere{};
	std::shared_ptr<AccGrid> grid; // Arc<AccGrid>
	static Geometry Plane_(Plane p) { return {RMD_GEOM_PLANE, p, {}, nullptr}; }
	static Geometry Sphere_(Sphere s) { return {RMD_GEOM_SPHERE, {}, s, nullptr}; }
	static Geometry Grid(std::shared_ptr<AccGrid> g) { return {RMD_GEOM_GRID, {}, {}, std::move(g)}; }
};
struct Object { // core/src/scene.rs:33-37
	Geometry geometry;
	Material material;
};
struct Scene { // core/src/scene.rs:42-52
	std::vector<Object> objects;
};

struct Transform { // src/transform.rs:4-14
	Vector3 position{0, 0, 0};
	static Transform identity() { return {}; }
};
struct CameraSettings { // src/trace.rs:32-40
	size_t backbuffer_width = 0, backbuffer_height = 0;
	double fov_vert = 55.0;
	Transform transform;
	double focal_length = 2.5, aperture_radius = 0.0;
};
struct Settings { // src/trace.rs:42-55 (+ the RNG seed the reference lacks)
	size_t worker_count = 1; // number of GPUs (the reference: num_cpus::get() threads)
	CameraSettings camera_settings;
	size_t sample_count = 1;
	size_t samples_per_iteration = 0;
	std::pair<size_t, size_t> tile_size{32, 32};
	size_t bounce_limit = 5;
	uint64_t seed = 0x5EED0001ull;
	bool use_dof = false; // opt-in: generate_primary_ray_with_dof (src/trace.rs:335-360); the reference's loop never calls it (:199)
	// opt-in (raymond_hip.h: RMD_RENDER_END_BLACK_PATHS): end zero-throughput paths in scenes with meshes too; false = reference-identical
	bool end_black_paths = false;
};

// core/src/tile.rs:13 `data: Vec<Vector3>` — the running sums of a tile, width * height of them, row-major.  Here a VIEW: the tiles of one
// message batch (a progressive pass's TileProgressed snapshots, or the TileFinished tiles) share the one page-locked block their pixels were
// downloaded into (rmd_framebuffer_download_tiles: the block is in Tile.data layout already), kept alive by its last tile — where the
// reference clones 24 KB per tile and message (src/trace.rs:212,218).  A tile that was not produced by a download owns its block.
class TileData {
  public:
	TileData() = default;
	explicit TileData(size_t n) : block_(new Vector3[n](), std::default_delete<Vector3[]>()), p_(static_cast<Vector3 *>(block_.get())), n_(n) {}
	TileData(std::shared_ptr<void> block, Vector3 *first, size_t n) : block_(std::move(block)), p_(first), n_(n) {}
	size_t size() const { return n_; }
	bool empty() const { return n_ == 0; }
	const Vector3 &operator[](size_t i) const { return p_[i]; }
	Vector3 &operator[](size_t i) { return p_[i]; }
	const Vector3 *data() const { return p_; }
	Vector3 *data() { return p_; }
	const Vector3 *begin() const { return p_; }
	const Vector3 *end() const { return p_ + n_; }
	void push_back(const Vector3 &v) { // (tests build small tiles by hand)
		TileData grown(n_ + 1);
		for (size_t i = 0; i < n_; i++) grown[i] = p_[i];
		grown[n_] = v;
		*this = std::move(grown);
	}

  private:
	std::shared_ptr<void> block_;
	Vector3 *p_ = nullptr;
	size_t n_ = 0;
};
struct Tile { // core/src/tile.rs:7-14
	size_t sample_count = 0, width = 0, height = 0, left = 0, top = 0;
	TileData data; // running sums, width*height.  EMPTY while the tile waits in the queue with its sums resident on a GPU (`resident`)
	int resident = -1; // the worker (GPU) whose device framebuffer holds the tile's sums; -1: `data` does (an extension: the reference's tiles live in RAM)
};
struct Message { // src/trace.rs:62-66
	enum Kind { TileFinished, TileProgressed } kind;
	Tile tile;
};

// src/trace.rs:70-135
class TaskHandle {
  public:
	using TileCallback = std::function<void(const Tile &)>;
	Settings settings;
	void set_callback(TileCallback cb) { callback_ = std::move(cb); }
	// Blocks until every worker is done, then assembles W*H radiance values (tile sums / sample_count), row-major (:82-113)
	std::vector<Vector3> await();
	std::optional<Message> poll();        // :115-117
	void async_await();                   // :119-134: drains leading TileProgressed messages into the callback
	bool finished() const;                // extension: alive_thread_count == 0 (what await() polls for, :89)
	double setup_seconds() const;         // extension (measurement): the longest a worker took to get ready — context, scene upload, framebuffer
	~TaskHandle();
	TaskHandle(TaskHandle &&) = default;
	struct Shared; // queue + channel shared with the workers (implementation detail)

  private:
	friend TaskHandle render_tiled(const Scene &, const Settings &);
	TaskHandle() = default;
	std::shared_ptr<Shared> shared_;
	std::vector<std::thread> workers_;
	TileCallback callback_;
};

// src/trace.rs:137-230
TaskHandle render_tiled(const Scene &scene, const Settings &settings);

// Tile generation of render_tiled (:142-173): column-major, edge tiles clamped
std::vector<rmd_tile_rect> generate_tiles(size_t width, size_t height, std::pair<size_t, size_t> tile_size);

// cli_old/src/main.rs:155-181 on the host: c = (1 - exp(-p * exposure))^(1/gamma); u8 = trunc(c * 255)
std::vector<uint8_t> tone_map(const std::vector<Vector3> &image, double exposure = 1.0, double gamma = 2.2);
void write_ppm(const std::string &path, const std::vector<uint8_t> &rgb8, size_t width, size_t height);

// core/src/project.rs:13-57 — the scene file: serde-JSON, enums externally tagged ({"Plane":{..}} | {"Sphere":{..}} |
// {"Mesh":"file.ply"}; {"Diffuse":[V,r]} | {"Metal":[V,r]} | {"Emission":[V,V,f,f]}), Vector3 as {"x","y","z"} or [x,y,z].
struct ProjectObject {
	enum Kind { PlaneGeometry, SphereGeometry, MeshGeometry } kind = PlaneGeometry;
	Plane plane{};
	Sphere sphere{};
	std::string mesh_path; // Geometry::Mesh(PathBuf)
	Material material{};
};
struct Project {
	std::vector<ProjectObject> objects;
	std::string base_dir; // directory of the project file: relative mesh paths are resolved against it (as raymond_amd/project.py does)
	static Project load(const std::string &path);                                       // project.rs:33-36
	static Project loads(const std::string &json_text, const std::string &base_dir = ""); // serde_json::from_str
	std::string dumps() const;                                                           // serde_json::to_string
	Scene build_scene() const; // project.rs:38-57: meshes through Mesh::load_ply + AccGrid::build_from_mesh
};
// server/src/protocol.rs:9-14: {"type":"TileProgressed"|"TileFinished","data":{sample_count,width,height,left,top,data:[V..]}}
std::string message_to_json(const Message &message);

// Benchmark inputs, identical to raymond_amd/scenes.py
Scene reflective_spheres();
Mesh lumpy_sphere_mesh(int n = 91, Vector3 extent = {2.3, 1.7, 1.0}, Vector3 centre = {0.0, 0.15, 0.0});
Scene gold_dragon_standin(int n = 91);

} // namespace raymond
