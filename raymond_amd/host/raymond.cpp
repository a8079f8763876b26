// Implementation of raymond.hpp (host side above the C-ABI).  Host code only: every pixel is produced by
// rmd_render_tiles; nothing here evaluates a ray.
#include "raymond.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

namespace raymond {

namespace {
void check(rmd_status s, rmd_context *ctx, const char *what) {
	if (s != RMD_OK) {
		const char *text = rmd_last_error(ctx);
		throw Error(s, std::string(what) + ": " + (text ? text : ""));
	}
}
} // namespace

// ---------------------------------------------------------------- Mesh (core/src/geometry/mesh.rs)
Mesh Mesh::load_ply(const std::string &path) {
	std::ifstream in(path);
	if (!in) throw Error(RMD_ERR_INVALID_ARGUMENT, "load_ply: cannot open " + path); // reference: unwrap() panic
	std::string line;
	size_t n_vertices = 0;
	// header (:66-77): only `element vertex N` matters
	while (std::getline(in, line)) {
		std::istringstream tok(line);
		std::string a, b;
		if (!(tok >> a)) throw Error(RMD_ERR_INVALID_ARGUMENT, "load_ply: empty header line"); // tokens.next().unwrap()
		if (a == "element") {
			tok >> b;
			if (b == "vertex") tok >> n_vertices;
		} else if (a == "end_header") {
			break;
		}
	}
	struct V {
		double p[3], n[3];
	};
	std::vector<V> verts;
	verts.reserve(n_vertices);
	for (size_t i = 0; i < n_vertices; i++) { // :80-90
		if (!std::getline(in, line)) throw Error(RMD_ERR_INVALID_ARGUMENT, "load_ply: vertex list truncated");
		std::istringstream tok(line);
		std::vector<double> v;
		double x;
		while (tok >> x) v.push_back(x);
		if (v.size() < 6) throw Error(RMD_ERR_INVALID_ARGUMENT, "load_ply: vertex line with fewer than 6 values");
		verts.push_back(V{{v[0], v[1], v[2]}, {v[3], v[4], v[5]}}); // uv (:87) and tangent (:88,:101-108) are not on the hot path
	}
	Mesh m;
	while (std::getline(in, line)) { // :93-118
		std::istringstream tok(line);
		std::vector<unsigned long> v;
		unsigned long x;
		while (tok >> x) v.push_back(x);
		if (v.empty()) throw Error(RMD_ERR_INVALID_ARGUMENT, "load_ply: empty face line"); // values[0] panics
		if (v[0] != 3) continue;                                                            // non-triangles are dropped (:116)
		if (v.size() < 4) throw Error(RMD_ERR_INVALID_ARGUMENT, "load_ply: triangle with fewer than 3 indices");
		for (int k = 1; k <= 3; k++) {
			if (v[k] >= verts.size()) throw Error(RMD_ERR_INVALID_ARGUMENT, "load_ply: vertex index out of range");
			for (int a = 0; a < 3; a++) m.tri_pos.push_back(verts[v[k]].p[a]);
		}
		for (int k = 1; k <= 3; k++)
			for (int a = 0; a < 3; a++) m.tri_nrm.push_back(verts[v[k]].n[a]);
	}
	return m;
}

void Mesh::bake_transform(Vector3 t) {
	for (size_t i = 0; i < tri_pos.size(); i++) tri_pos[i] += t[i % 3];
}

// ---------------------------------------------------------------- AccGrid
std::shared_ptr<AccGrid> AccGrid::build_from_mesh(const Mesh &mesh) {
	std::shared_ptr<AccGrid> g(new AccGrid());
	check(rmd_grid_build_from_mesh(mesh.tri_pos.data(), mesh.tri_nrm.data(), mesh.triangle_count(), &g->build_), nullptr, "AccGrid::build_from_mesh");
	check(rmd_grid_build_describe(g->build_, &g->desc_), nullptr, "AccGrid::build_from_mesh");
	return g;
}
AccGrid::~AccGrid() { rmd_grid_build_destroy(build_); }

// ---------------------------------------------------------------- render_tiled
std::vector<rmd_tile_rect> generate_tiles(size_t width, size_t height, std::pair<size_t, size_t> ts) {
	std::vector<rmd_tile_rect> tiles;
	size_t x = 0, y = 0;
	for (;;) {
		size_t x1 = std::min(x + ts.first, width), y1 = std::min(y + ts.second, height);
		tiles.push_back(rmd_tile_rect{(uint32_t)x, (uint32_t)y, (uint32_t)(x1 - x), (uint32_t)(y1 - y)});
		y += ts.second;
		if (y >= height) {
			y = 0;
			x += ts.first;
		}
		if (x >= width) break;
	}
	return tiles;
}

struct TaskHandle::Shared {
	std::mutex m;
	std::condition_variable cv;
	std::deque<Tile> queue;      // MsQueue<Tile> (:138)
	std::deque<Message> channel; // mpsc::channel (:139)
	size_t alive = 0;            // alive_thread_count (:175)
	size_t in_flight = 0;        // tiles popped by a worker and not yet finished or re-queued
	std::string error;
	double setup_s = 0.0;        // measurement: the longest a worker took to get ready (context, scene upload, framebuffer)
};

namespace {

void flatten(const Scene &scene, std::vector<rmd_object> &objs, std::vector<rmd_grid_desc> &grids) {
	std::vector<const AccGrid *> seen;
	for (const Object &o : scene.objects) {
		rmd_object r;
		std::memset(&r, 0, sizeof(r));
		r.geometry_kind = o.geometry.kind;
		if (o.geometry.kind == RMD_GEOM_PLANE) {
			for (int a = 0; a < 3; a++) r.origin[a] = o.geometry.plane.origin[a], r.normal[a] = o.geometry.plane.normal[a];
		} else if (o.geometry.kind == RMD_GEOM_SPHERE) {
			for (int a = 0; a < 3; a++) r.origin[a] = o.geometry.sphere.origin[a];
			r.radius = o.geometry.sphere.radius;
		} else {
			const AccGrid *g = o.geometry.grid.get();
			size_t gi = std::find(seen.begin(), seen.end(), g) - seen.begin();
			if (gi == seen.size()) {
				seen.push_back(g);
				grids.push_back(g->desc());
			}
			r.grid_index = (uint32_t)gi;
		}
		r.material.kind = o.material.kind;
		for (int a = 0; a < 3; a++) r.material.color[a] = o.material.color[a];
		r.material.roughness = o.material.roughness;
		for (int a = 0; a < 5; a++) r.material.emission_aux[a] = o.material.aux[a];
		objs.push_back(r);
	}
}

// Page-locked blocks for the downloads, recycled: a block goes back to the pool when the last tile (message) that views it is dropped.
struct BlockPool : std::enable_shared_from_this<BlockPool> {
	static std::shared_ptr<BlockPool> shared() {
		static std::shared_ptr<BlockPool> pool = std::make_shared<BlockPool>();
		return pool;
	}
	std::mutex m;
	std::vector<std::pair<void *, size_t>> free_blocks;
	// (the blocks of the process-wide pool are left to the operating system at exit: the HIP runtime may be gone by then)
	std::shared_ptr<void> get(rmd_context *ctx, size_t bytes) {
		void *p = nullptr;
		size_t have = 0;
		{
			std::lock_guard<std::mutex> lock(m);
			for (size_t i = 0; i < free_blocks.size(); i++)
				if (free_blocks[i].second >= bytes) {
					p = free_blocks[i].first, have = free_blocks[i].second;
					free_blocks.erase(free_blocks.begin() + (long)i);
					break;
				}
		}
		if (!p) {
			check(rmd_host_alloc(ctx, bytes, &p), ctx, "rmd_host_alloc");
			have = bytes;
		}
		std::shared_ptr<BlockPool> self = shared_from_this();
		return std::shared_ptr<void>(p, [self, have](void *q) {
			std::lock_guard<std::mutex> lock(self->m);
			if (self->free_blocks.size() < 4) self->free_blocks.emplace_back(q, have);
			else rmd_host_free(nullptr, q);
		});
	}
};

// One worker = one GPU.  Pops a batch of tiles, adds `step` samples to each with ONE rmd_render_tiles call per sample count, then reports them
// finished or re-queues them (src/trace.rs:188-221).  The tiles' sums stay in this GPU's framebuffer; only what a message carries is downloaded,
// on the copy stream, while the next batch renders (the messages of batch k are sent while batch k + 1 runs).
void worker_main(std::shared_ptr<TaskHandle::Shared> sh, int device, int me, size_t workers, Scene scene, Settings st, size_t batch) {
	rmd_context *ctx = nullptr;
	rmd_scene *dscene = nullptr;
	double *fb = nullptr;
	const size_t W = st.camera_settings.backbuffer_width, H = st.camera_settings.backbuffer_height;
	// a batch whose download is on its way: the tiles that become messages, in message order
	struct Pending {
		std::vector<Message> messages;
		std::vector<Tile> requeue; // (several workers: tiles that go back to the queue with their sums in RAM)
		size_t taken = 0;          // tiles of the batch that were neither finished nor re-queued at once (they leave `in_flight` now)
	};
	std::optional<Pending> pending;
	auto flush = [&]() { // the previous batch's download has to arrive before its messages can be sent
		if (!pending) return;
		check(rmd_context_wait_transfers(ctx), ctx, "rmd_context_wait_transfers");
		std::lock_guard<std::mutex> lock(sh->m);
		for (Message &m : pending->messages) sh->channel.push_back(std::move(m));
		for (Tile &t : pending->requeue) sh->queue.push_back(std::move(t));
		sh->in_flight -= pending->taken;
		pending.reset();
		sh->cv.notify_all();
	};
	try {
		const auto t_setup = std::chrono::steady_clock::now();
		check(rmd_context_create(device, &ctx), nullptr, "rmd_context_create");
		std::vector<rmd_object> objs;
		std::vector<rmd_grid_desc> grids;
		flatten(scene, objs, grids);
		check(rmd_scene_create(ctx, objs.data(), (uint32_t)objs.size(), grids.data(), (uint32_t)grids.size(), &dscene), ctx, "rmd_scene_create");
		check(rmd_framebuffer_alloc(ctx, (uint32_t)W, (uint32_t)H, &fb), ctx, "rmd_framebuffer_alloc"); // zeroed: a fresh tile's sums
		{
			std::lock_guard<std::mutex> lock(sh->m);
			sh->setup_s = std::max(sh->setup_s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup).count());
		}
		rmd_camera cam;
		std::memset(&cam, 0, sizeof(cam));
		cam.backbuffer_width = (uint32_t)W, cam.backbuffer_height = (uint32_t)H, cam.fov_vert = st.camera_settings.fov_vert;
		for (int a = 0; a < 3; a++) cam.position[a] = st.camera_settings.transform.position[a];
		cam.focal_length = st.camera_settings.focal_length, cam.aperture_radius = st.camera_settings.aperture_radius;
		const uint32_t flags = (st.use_dof ? RMD_RENDER_DOF : 0u) | (st.end_black_paths ? RMD_RENDER_END_BLACK_PATHS : 0u); // 0 = the reference's loop: pinhole (:199), every sample identical
		const size_t step = st.samples_per_iteration ? st.samples_per_iteration : st.sample_count;
		std::shared_ptr<BlockPool> pool = BlockPool::shared(); // process-wide: page-locking 50 MB costs about as much as moving them
		for (;;) {
			std::vector<Tile> mine;
			{
				// try_pop (:189).  The reference's worker leaves as soon as the queue is empty (:191-194); with one GPU call
				// per batch a queue that is only momentarily empty — the other workers hold every tile and will re-queue them
				// for the next progressive pass — would collapse the pool to one GPU, so a worker leaves only when no tile is
				// queued AND none is in flight.
				std::unique_lock<std::mutex> lock(sh->m);
				if (sh->queue.empty() && pending) { // nothing to start: send what is pending (it may re-queue tiles or end the render)
					lock.unlock();
					flush();
					lock.lock();
				}
				sh->cv.wait(lock, [&] { return !sh->queue.empty() || sh->in_flight == 0 || !sh->error.empty(); });
				while (sh->error.empty() && !sh->queue.empty() && mine.size() < batch) {
					mine.push_back(std::move(sh->queue.front()));
					sh->queue.pop_front();
				}
				sh->in_flight += mine.size();
			}
			if (mine.empty()) break;
			// tiles that arrive with their sums in RAM (another GPU rendered their earlier passes): into this GPU's framebuffer
			{
				std::vector<rmd_tile_rect> rects;
				std::vector<double> packed;
				for (Tile &t : mine)
					if (t.resident != me && t.sample_count != 0) {
						rects.push_back(rmd_tile_rect{(uint32_t)t.left, (uint32_t)t.top, (uint32_t)t.width, (uint32_t)t.height});
						const double *src = reinterpret_cast<const double *>(t.data.data());
						packed.insert(packed.end(), src, src + t.data.size() * 3);
					}
				if (!rects.empty()) check(rmd_framebuffer_upload_tiles(ctx, packed.data(), fb, (uint32_t)W, (uint32_t)H, rects.data(), (uint32_t)rects.size()), ctx, "rmd_framebuffer_upload_tiles");
			}
			// tiles of one batch may be at different sample counts: one launch per count (enqueued, not waited for)
			std::map<size_t, std::vector<size_t>> by_count;
			for (size_t i = 0; i < mine.size(); i++) by_count[mine[i].sample_count].push_back(i);
			for (auto &grp : by_count) {
				const size_t begin = grp.first, n = std::min(step, st.sample_count - begin);
				std::vector<rmd_tile_rect> rects;
				for (size_t i : grp.second) rects.push_back(rmd_tile_rect{(uint32_t)mine[i].left, (uint32_t)mine[i].top, (uint32_t)mine[i].width, (uint32_t)mine[i].height});
				rmd_settings rs;
				std::memset(&rs, 0, sizeof(rs));
				rs.bounce_limit = (uint32_t)st.bounce_limit, rs.sample_begin = (uint32_t)begin, rs.sample_count = (uint32_t)n, rs.seed = st.seed, rs.flags = flags;
				check(rmd_render_tiles_async(ctx, dscene, &cam, &rs, rects.data(), (uint32_t)rects.size(), fb), ctx, "rmd_render_tiles");
				for (size_t i : grp.second) mine[i].sample_count += n, mine[i].resident = me, mine[i].data = TileData(); // :207 — the sums are on this GPU now
			}
			// the previous batch's messages go out while this batch renders
			flush();
			// what of this batch has to come to the host: finished tiles (:211-212), progress snapshots (:217-219), and — with several GPUs — every
			// tile that goes back to the shared queue (another GPU may take it next)
			Pending next;
			std::vector<rmd_tile_rect> rects;
			std::vector<size_t> want;
			std::vector<Tile> resident_requeue;
			size_t pixels = 0;
			for (size_t i = 0; i < mine.size(); i++) {
				Tile &t = mine[i];
				const bool finished = t.sample_count == st.sample_count;
				const bool progressed = !finished && st.samples_per_iteration != 0 && t.sample_count % st.samples_per_iteration == 0;
				if (finished || progressed || workers > 1) {
					want.push_back(i);
					rects.push_back(rmd_tile_rect{(uint32_t)t.left, (uint32_t)t.top, (uint32_t)t.width, (uint32_t)t.height});
					pixels += t.width * t.height;
				}
			}
			if (!want.empty()) {
				std::shared_ptr<void> block = pool->get(ctx, pixels * 24);
				check(rmd_framebuffer_download_tiles_async(ctx, fb, (uint32_t)W, (uint32_t)H, rects.data(), (uint32_t)rects.size(), static_cast<double *>(block.get())), ctx, "rmd_framebuffer_download_tiles");
				Vector3 *p = static_cast<Vector3 *>(block.get());
				for (size_t i : want) {
					mine[i].data = TileData(block, p, mine[i].width * mine[i].height);
					p += mine[i].width * mine[i].height;
				}
			}
			for (Tile &t : mine) {
				const bool finished = t.sample_count == st.sample_count;
				const bool progressed = !finished && st.samples_per_iteration != 0 && t.sample_count % st.samples_per_iteration == 0;
				if (finished) {
					t.resident = -1;
					next.messages.push_back(Message{Message::TileFinished, std::move(t)});
					next.taken++;
				} else if (workers > 1) { // back to the shared queue with its sums in RAM, once they have arrived
					t.resident = -1;
					if (progressed) next.messages.push_back(Message{Message::TileProgressed, t});
					next.requeue.push_back(std::move(t));
					next.taken++;
				} else { // one GPU: the tile goes back to the queue at once, sums resident; its snapshot follows when it has arrived
					if (progressed) {
						Tile snapshot = t;
						snapshot.resident = -1;
						next.messages.push_back(Message{Message::TileProgressed, std::move(snapshot)});
					}
					t.data = TileData();
					resident_requeue.push_back(std::move(t));
				}
			}
			{
				std::lock_guard<std::mutex> lock(sh->m);
				sh->in_flight -= resident_requeue.size();
				for (Tile &t : resident_requeue) sh->queue.push_back(std::move(t));
				sh->cv.notify_all();
			}
			pending = std::move(next);
		}
		flush();
		check(rmd_context_synchronize(ctx), ctx, "rmd_context_synchronize"); // a device fault of the last launch surfaces here at the latest
	} catch (const std::exception &e) {
		std::lock_guard<std::mutex> lock(sh->m);
		if (sh->error.empty()) sh->error = e.what(); // the waiting workers see it and leave
		sh->cv.notify_all();
	}
	if (fb) rmd_framebuffer_free(ctx, fb);
	rmd_scene_destroy(dscene);
	rmd_context_destroy(ctx);
	std::lock_guard<std::mutex> lock(sh->m);
	sh->alive--; // :192
	sh->cv.notify_all();
}

} // namespace

TaskHandle render_tiled(const Scene &scene, const Settings &settings) {
	TaskHandle h;
	h.settings = settings;
	h.shared_ = std::make_shared<TaskHandle::Shared>();
	const CameraSettings &cam = settings.camera_settings;
	for (const rmd_tile_rect &r : generate_tiles(cam.backbuffer_width, cam.backbuffer_height, settings.tile_size)) {
		Tile t;
		t.left = r.left, t.top = r.top, t.width = r.width, t.height = r.height;
		// (no data: a fresh tile's sums are the zeros of its worker's device framebuffer)
		h.shared_->queue.push_back(std::move(t));
	}
	const size_t workers = std::max<size_t>(1, settings.worker_count);
	const size_t batch = std::max<size_t>(1, (h.shared_->queue.size() + workers * 4 - 1) / (workers * 4));
	h.shared_->alive = workers;
	// worker w drives GPU w; RAYMOND_REHEARSE_ON_DEVICE0=1 (tests on a one-GPU box) gives every worker its own context on GPU 0
	const bool rehearse = std::getenv("RAYMOND_REHEARSE_ON_DEVICE0") != nullptr;
	for (size_t w = 0; w < workers; w++)
		h.workers_.emplace_back(worker_main, h.shared_, rehearse ? 0 : (int)w, (int)w, workers, scene, settings, workers == 1 ? h.shared_->queue.size() : batch);
	return h;
}

TaskHandle::~TaskHandle() {
	for (std::thread &t : workers_)
		if (t.joinable()) t.join();
}

std::vector<Vector3> TaskHandle::await() {
	const CameraSettings &cam = settings.camera_settings;
	std::vector<Vector3> out(cam.backbuffer_width * cam.backbuffer_height, Vector3{0, 0, 0});
	std::unique_lock<std::mutex> lock(shared_->m);
	shared_->cv.wait(lock, [&] { return shared_->alive == 0; }); // the reference polls every 500 ms (:88-110)
	if (!shared_->error.empty()) throw Error(RMD_ERR_HIP, shared_->error); // the reference would hang after a worker panic
	while (!shared_->channel.empty()) {
		Message m = std::move(shared_->channel.front());
		shared_->channel.pop_front();
		if (m.kind != Message::TileFinished) break; // :101-103
		const Tile &t = m.tile;
		for (size_t y = 0; y < t.height; y++)
			for (size_t x = 0; x < t.width; x++) {
				Vector3 s = t.data[x + y * t.width];
				for (double &c : s) c /= (double)t.sample_count; // :95
				out[x + t.left + (y + t.top) * cam.backbuffer_width] = s;
			}
	}
	return out;
}

double TaskHandle::setup_seconds() const {
	std::lock_guard<std::mutex> lock(shared_->m);
	return shared_->setup_s;
}

bool TaskHandle::finished() const {
	std::lock_guard<std::mutex> lock(shared_->m);
	return shared_->alive == 0;
}

std::optional<Message> TaskHandle::poll() {
	std::lock_guard<std::mutex> lock(shared_->m);
	if (shared_->channel.empty()) return std::nullopt;
	Message m = std::move(shared_->channel.front());
	shared_->channel.pop_front();
	return m;
}

void TaskHandle::async_await() {
	for (;;) {
		std::optional<Message> m;
		{
			std::lock_guard<std::mutex> lock(shared_->m);
			if (shared_->channel.empty() || shared_->channel.front().kind != Message::TileProgressed) return;
			m = std::move(shared_->channel.front());
			shared_->channel.pop_front();
		}
		if (callback_) callback_(m->tile);
	}
}

// ---------------------------------------------------------------- output stage (cli_old/src/main.rs:155-197)
std::vector<uint8_t> tone_map(const std::vector<Vector3> &image, double exposure, double gamma) {
	std::vector<uint8_t> out(image.size() * 3, 0);
	for (size_t i = 0; i < image.size(); i++) {
		double v[3];
		bool ok = true;
		for (int c = 0; c < 3; c++) {
			double tm = 1.0 - std::exp(image[i][c] * -1.0 * exposure);
			tm = std::pow(tm, 1.0 / gamma);
			v[c] = tm * 255.0;
			ok = ok && v[c] > -1.0 && v[c] < 256.0;
		}
		if (ok)
			for (int c = 0; c < 3; c++) out[i * 3 + c] = (uint8_t)v[c];
	}
	return out;
}

void write_ppm(const std::string &path, const std::vector<uint8_t> &rgb8, size_t width, size_t height) {
	std::ofstream f(path, std::ios::binary);
	f << "P6\n" << width << " " << height << "\n255\n";
	f.write(reinterpret_cast<const char *>(rgb8.data()), (std::streamsize)rgb8.size());
}

// ---------------------------------------------------------------- benchmark inputs (mirror of raymond_amd/scenes.py)
namespace {
void room_planes(Scene &s) { // cli_old/src/main.rs:77-127
	s.objects.push_back({Geometry::Plane_({{0, -1, 0}, {0, 1, 0}}), Material::Diffuse({0.75, 0.75, 0.75}, 0.5)});
	s.objects.push_back({Geometry::Plane_({{0, 2, 0}, {0, -1, 0}}), Material::Emission({1.5, 1.5, 1.5}, {1, 1, 1}, 0.27, 0.0)});
	s.objects.push_back({Geometry::Plane_({{0, 0, -2}, {0, 0, 1}}), Material::Diffuse({1, 1, 1}, 0.4)});
	s.objects.push_back({Geometry::Plane_({{0, 0, 5}, {0, 0, -1}}), Material::Diffuse({0, 0, 0}, 0.9)});
	s.objects.push_back({Geometry::Plane_({{-2, 0, 0}, {1, 0, 0}}), Material::Diffuse({0, 0, 0}, 0.3)});
	s.objects.push_back({Geometry::Plane_({{2, 0, 0}, {-1, 0, 0}}), Material::Diffuse({0, 0, 0}, 0.3)});
}
} // namespace

Scene reflective_spheres() {
	Scene s;
	s.objects.push_back({Geometry::Sphere_({{-1.0, -0.5, 3.5}, 0.5}), Material::Diffuse({1.0, 0.0, 0.0}, 0.02)});
	s.objects.push_back({Geometry::Sphere_({{0.74, -0.25, 3.5}, 0.75}), Material::Metal({0.05, 0.25, 1.0}, 0.01)});
	room_planes(s);
	return s;
}

// Same arithmetic, in the same order, as raymond_amd.scenes.lumpy_sphere_mesh (IEEE-exact operations only),
// so both produce bit-identical triangles.
Mesh lumpy_sphere_mesh(int n, Vector3 extent, Vector3 centre) {
	std::map<std::tuple<int, int, int>, int> index;
	std::vector<std::array<int, 3>> lattice;
	auto vid = [&](int i, int j, int k) {
		auto key = std::make_tuple(i, j, k);
		auto it = index.find(key);
		if (it != index.end()) return it->second;
		int v = (int)lattice.size();
		index.emplace(key, v);
		lattice.push_back({i, j, k});
		return v;
	};
	std::vector<std::array<int, 3>> faces;
	const int spec[6][4] = {{0, n, 1, 2}, {0, 0, 2, 1}, {1, n, 2, 0}, {1, 0, 0, 2}, {2, n, 0, 1}, {2, 0, 1, 0}};
	for (const auto &f : spec)
		for (int a = 0; a < n; a++)
			for (int b = 0; b < n; b++) {
				auto p = [&](int da, int db) {
					int c[3] = {0, 0, 0};
					c[f[0]] = f[1], c[f[2]] = a + da, c[f[3]] = b + db;
					return vid(c[0], c[1], c[2]);
				};
				int v00 = p(0, 0), v10 = p(1, 0), v11 = p(1, 1), v01 = p(0, 1);
				faces.push_back({v00, v10, v11});
				faces.push_back({v00, v11, v01});
			}
	const size_t nv = lattice.size();
	std::vector<Vector3> p(nv);
	auto t3 = [](double t) { return (4.0 * t * t - 3.0) * t; };
	auto t2 = [](double t) { return 2.0 * t * t - 1.0; };
	const double step = 2.0 / n;
	Vector3 half{0, 0, 0};
	for (size_t i = 0; i < nv; i++) {
		double c[3], d[3];
		for (int a = 0; a < 3; a++) c[a] = (double)lattice[i][a] * step - 1.0;
		double len = std::sqrt((c[0] * c[0] + c[1] * c[1]) + c[2] * c[2]);
		for (int a = 0; a < 3; a++) d[a] = c[a] / len;
		double radius = 1.0 + 0.22 * t3(d[0]) * t3(d[1]) + 0.15 * t2(d[2]) * t3(d[1]) + 0.10 * t3(d[2]) * t2(d[0]);
		for (int a = 0; a < 3; a++) {
			p[i][a] = d[a] * radius;
			half[a] = std::max(half[a], std::fabs(p[i][a]));
		}
	}
	Vector3 scale;
	for (int a = 0; a < 3; a++) scale[a] = extent[a] * 0.5 / half[a];
	for (size_t i = 0; i < nv; i++)
		for (int a = 0; a < 3; a++) p[i][a] = p[i][a] * scale[a] + centre[a];
	std::vector<Vector3> fn(faces.size()), vn(nv, Vector3{0, 0, 0});
	for (size_t f = 0; f < faces.size(); f++) {
		const Vector3 &p0 = p[faces[f][0]], &p1 = p[faces[f][1]], &p2 = p[faces[f][2]];
		double e1[3], e2[3];
		for (int a = 0; a < 3; a++) e1[a] = p1[a] - p0[a], e2[a] = p2[a] - p0[a];
		fn[f] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
	}
	for (int k = 0; k < 3; k++) // np.add.at(vn, faces[:, k], fn): sequential, corner by corner
		for (size_t f = 0; f < faces.size(); f++)
			for (int a = 0; a < 3; a++) vn[faces[f][k]][a] += fn[f][a];
	for (size_t i = 0; i < nv; i++) {
		double len = std::sqrt((vn[i][0] * vn[i][0] + vn[i][1] * vn[i][1]) + vn[i][2] * vn[i][2]);
		for (int a = 0; a < 3; a++) vn[i][a] = vn[i][a] / len;
	}
	Mesh m;
	m.tri_pos.reserve(faces.size() * 9), m.tri_nrm.reserve(faces.size() * 9);
	for (const auto &f : faces) {
		for (int k = 0; k < 3; k++)
			for (int a = 0; a < 3; a++) m.tri_pos.push_back(p[f[k]][a]);
		for (int k = 0; k < 3; k++)
			for (int a = 0; a < 3; a++) m.tri_nrm.push_back(vn[f[k]][a]);
	}
	return m;
}

Scene gold_dragon_standin(int n) {
	Mesh mesh = lumpy_sphere_mesh(n);
	mesh.bake_transform({0.0, -0.3, 2.9}); // cli_old/src/main.rs:61
	Scene s;
	s.objects.push_back({Geometry::Sphere_({{-1.0, -0.5, 3.5}, 0.5}), Material::Diffuse({1.0, 0.0, 0.0}, 0.02)});
	s.objects.push_back({Geometry::Grid(AccGrid::build_from_mesh(mesh)), Material::Metal({1.0, 1.0, 0.1}, 0.15)}); // :63,:72-75
	room_planes(s);
	return s;
}

} // namespace raymond
