// raymond_cli — the executable caller of the host mirror; follows cli_old/src/main.rs:35-198
// (build scene -> Settings -> render_tiled -> await -> tone-map -> image file).
//
//   raymond_cli render <spheres|dragon[:n]> W H SPP BOUNCES out.ppm [--raw out.f64] [--gpus N] [--spi K] [--aperture R] [--end-black-paths 1]
//   raymond_cli mesh N out.bin            procedural stand-in mesh as raw f64 (tri_pos then tri_nrm)
//   raymond_cli ply in.ply out.bin        Mesh::load_ply + bake_transform(0,-0.3,2.9), raw f64 as above
//   raymond_cli tiles W H TW TH           tile generation order of render_tiled, one "left top width height" per line
//   raymond_cli project in.json dump|json Project::load (core/src/project.rs): flattened scene, or the re-serialised JSON
//   raymond_cli tilemsg W H               a TileFinished message in the wire form of server/src/protocol.rs
//   raymond_cli hostapi <spheres|dragon[:n]> W H SPP BOUNCES SPI [--end-black-paths 1]
//                                         what a drop-in caller gets: one JSON line with the wall time of render_tiled -> last TileFinished
//   (render also accepts `project:in.json` as its scene)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>

#include "raymond.hpp"

using namespace raymond;

// A consumer of a progressive render: every message through poll() (src/trace.rs:115-117) as it arrives — TileProgressed snapshots are counted,
// TileFinished tiles are assembled into the image as await() would (:93-99).  (await() itself stops collecting at the first message that is not
// TileFinished, :101-103: with snapshots of several workers in the channel it is only safe once they have been drained.)
static std::vector<Vector3> consume(TaskHandle &handle, const Settings &st, size_t &progressed, double *last_finished_s = nullptr,
                                    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now()) {
	const size_t W = st.camera_settings.backbuffer_width, H = st.camera_settings.backbuffer_height;
	const size_t n_tiles = generate_tiles(W, H, st.tile_size).size();
	std::vector<Tile> finished_tiles; // (zero-copy views of the download's block: taking them is cheap; the image is assembled afterwards)
	finished_tiles.reserve(n_tiles);
	for (;;) {
		const bool done = handle.finished(); // read BEFORE the channel is drained: nothing is sent after the last worker has left
		while (std::optional<Message> m = handle.poll()) {
			if (m->kind == Message::TileProgressed) {
				progressed++;
				continue;
			}
			finished_tiles.push_back(std::move(m->tile));
			if (finished_tiles.size() == n_tiles && last_finished_s) *last_finished_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		}
		if (done || finished_tiles.size() == n_tiles) break;
		std::this_thread::sleep_for(std::chrono::microseconds(100));
	}
	std::vector<Vector3> image(W * H, Vector3{0, 0, 0});
	for (const Tile &t : finished_tiles)
		for (size_t y = 0; y < t.height; y++)
			for (size_t x = 0; x < t.width; x++) {
				Vector3 v = t.data[x + y * t.width];
				for (double &c : v) c /= (double)t.sample_count; // :95
				image[x + t.left + (y + t.top) * W] = v;
			}
	handle.await(); // waits for the workers to leave (they free their device memory after their last message); rethrows a worker's error
	return image;
}

static void dump_mesh(const Mesh &m, const std::string &path) {
	std::ofstream f(path, std::ios::binary);
	f.write(reinterpret_cast<const char *>(m.tri_pos.data()), (std::streamsize)(m.tri_pos.size() * 8));
	f.write(reinterpret_cast<const char *>(m.tri_nrm.data()), (std::streamsize)(m.tri_nrm.size() * 8));
}

int main(int argc, char **argv) {
	try {
		if (argc >= 4 && !std::strcmp(argv[1], "mesh")) {
			dump_mesh(lumpy_sphere_mesh(std::atoi(argv[2])), argv[3]);
			return 0;
		}
		if (argc >= 4 && !std::strcmp(argv[1], "ply")) {
			Mesh m = Mesh::load_ply(argv[2]);
			m.bake_transform({0.0, -0.3, 2.9});
			dump_mesh(m, argv[3]);
			std::printf("%zu triangles\n", m.triangle_count());
			return 0;
		}
		if (argc >= 4 && !std::strcmp(argv[1], "project")) {
			Project p = Project::load(argv[2]);
			if (!std::strcmp(argv[3], "json")) {
				std::printf("%s\n", p.dumps().c_str());
				return 0;
			}
			Scene sc = p.build_scene();
			for (const Object &o : sc.objects) {
				const Geometry &g = o.geometry;
				if (g.kind == RMD_GEOM_PLANE) std::printf("plane %.17g %.17g %.17g %.17g %.17g %.17g", g.plane.origin[0], g.plane.origin[1], g.plane.origin[2], g.plane.normal[0], g.plane.normal[1], g.plane.normal[2]);
				else if (g.kind == RMD_GEOM_SPHERE) std::printf("sphere %.17g %.17g %.17g %.17g", g.sphere.origin[0], g.sphere.origin[1], g.sphere.origin[2], g.sphere.radius);
				else {
					const rmd_grid_desc &d = g.grid->desc();
					unsigned long long h = 1469598103934665603ull; // FNV-1a over cells then mapping_table
					for (uint64_t i = 0; i < d.n_cells; i++) h = (h ^ d.cells[i]) * 1099511628211ull;
					for (uint64_t i = 0; i < d.n_mapping; i++) h = (h ^ d.mapping_table[i]) * 1099511628211ull;
					std::printf("grid %u %u %u %llu %llu %llu", d.resolution[0], d.resolution[1], d.resolution[2], (unsigned long long)d.n_tris, (unsigned long long)d.n_mapping, h);
				}
				const Material &m = o.material;
				std::printf(" | %u %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", m.kind, m.color[0], m.color[1], m.color[2], m.roughness, m.aux[0], m.aux[1], m.aux[2], m.aux[3], m.aux[4]);
			}
			return 0;
		}
		if (argc >= 4 && !std::strcmp(argv[1], "tilemsg")) {
			Message msg;
			msg.kind = Message::TileFinished;
			msg.tile.width = std::atoi(argv[2]), msg.tile.height = std::atoi(argv[3]), msg.tile.left = 32, msg.tile.top = 64, msg.tile.sample_count = 7;
			for (size_t i = 0; i < msg.tile.width * msg.tile.height; i++) msg.tile.data.push_back({0.125 * (double)i, 1.0 / (double)(i + 3), -2.5e-7 * (double)i});
			std::printf("%s\n", message_to_json(msg).c_str());
			return 0;
		}
		if (argc >= 6 && !std::strcmp(argv[1], "tiles")) {
			for (const rmd_tile_rect &t : generate_tiles(std::atoi(argv[2]), std::atoi(argv[3]), {std::atoi(argv[4]), std::atoi(argv[5])}))
				std::printf("%u %u %u %u\n", t.left, t.top, t.width, t.height);
			return 0;
		}
		if (argc >= 8 && !std::strcmp(argv[1], "hostapi")) {
			// The measurement bench.py's `host_api` block reports: the scene is built first (as cli_old does, main.rs:45-127), a tiny untimed render
			// initialises the process's HIP runtime, then ONE render_tiled call is timed from the call to the last TileFinished message, the
			// TileProgressed snapshots handed to a callback as they arrive (src/trace.rs:119-134).
			std::string what = argv[2];
			Settings st;
			st.camera_settings.backbuffer_width = std::atoi(argv[3]), st.camera_settings.backbuffer_height = std::atoi(argv[4]);
			st.camera_settings.fov_vert = 55.0, st.camera_settings.focal_length = 2.5;
			st.sample_count = std::atoi(argv[5]), st.bounce_limit = std::atoi(argv[6]), st.samples_per_iteration = std::atoi(argv[7]);
			st.tile_size = {32, 32};
			for (int i = 8; i + 1 < argc; i += 2)
				if (!std::strcmp(argv[i], "--end-black-paths")) st.end_black_paths = std::atoi(argv[i + 1]) != 0;
			Scene scene = what == "spheres" ? reflective_spheres() : gold_dragon_standin(what.size() > 7 ? std::atoi(what.c_str() + 7) : 91);
			{
				Settings warm = st;
				warm.camera_settings.backbuffer_width = warm.camera_settings.backbuffer_height = 64, warm.sample_count = 1, warm.samples_per_iteration = 0;
				render_tiled(reflective_spheres(), warm).await();
			}
			const int reps = 3;
			double best = 1e30, best_setup = 0.0, best_await = 0.0;
			size_t progressed = 0, finished_tiles = 0;
			double checksum = 0.0;
			for (int rep = 0; rep < reps; rep++) {
				progressed = 0;
				const auto t0 = std::chrono::steady_clock::now();
				TaskHandle handle = render_tiled(scene, st);
				// wall: call -> the last TileFinished message has ARRIVED at the consumer (every message taken through poll() as it comes); then the
				// image is assembled as await() would (:93-99) and the workers' teardown is waited for
				double secs = 0.0;
				std::vector<Vector3> image = consume(handle, st, progressed, &secs, t0);
				const double secs_await = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
				if (secs < best) best = secs, best_setup = handle.setup_seconds(), best_await = secs_await;
				checksum = 0.0;
				for (const Vector3 &v : image)
					for (double c : v)
						if (c == c) checksum += c;
				finished_tiles = generate_tiles(st.camera_settings.backbuffer_width, st.camera_settings.backbuffer_height, st.tile_size).size();
			}
			const double n = (double)st.camera_settings.backbuffer_width * st.camera_settings.backbuffer_height * st.sample_count;
			std::printf("{\"scene\": \"%s\", \"width\": %zu, \"height\": %zu, \"spp\": %zu, \"bounces\": %zu, \"samples_per_iteration\": %zu, \"wall_ms\": %.3f, "
			            "\"setup_ms\": %.3f, \"image_assembled_ms\": %.3f, \"msamples_per_s\": %.2f, \"msamples_per_s_after_setup\": %.2f, \"tile_progressed_messages\": %zu, \"tiles\": %zu, "
			            "\"mean_radiance_sum\": %.17g, \"runs\": %d}\n",
			            what.c_str(), st.camera_settings.backbuffer_width, st.camera_settings.backbuffer_height, st.sample_count, st.bounce_limit, st.samples_per_iteration,
			            best * 1e3, best_setup * 1e3, best_await * 1e3, n / best / 1e6, n / (best - best_setup) / 1e6, progressed, finished_tiles, checksum, reps);
			return 0;
		}
		if (argc >= 8 && !std::strcmp(argv[1], "render")) {
			const auto t0 = std::chrono::steady_clock::now(); // cli_old/src/main.rs:36
			std::string what = argv[2], raw;
			Settings st;
			st.camera_settings.backbuffer_width = std::atoi(argv[3]);
			st.camera_settings.backbuffer_height = std::atoi(argv[4]);
			st.camera_settings.fov_vert = 55.0, st.camera_settings.focal_length = 2.5; // :134-141
			st.sample_count = std::atoi(argv[5]);
			st.bounce_limit = std::atoi(argv[6]);
			st.tile_size = {32, 32}; // :147
			for (int i = 8; i + 1 < argc; i += 2) {
				if (!std::strcmp(argv[i], "--raw")) raw = argv[i + 1];
				else if (!std::strcmp(argv[i], "--gpus")) st.worker_count = std::atoi(argv[i + 1]);
				else if (!std::strcmp(argv[i], "--spi")) st.samples_per_iteration = std::atoi(argv[i + 1]);
				else if (!std::strcmp(argv[i], "--aperture")) st.camera_settings.aperture_radius = std::atof(argv[i + 1]), st.use_dof = true;
				else if (!std::strcmp(argv[i], "--end-black-paths")) st.end_black_paths = std::atoi(argv[i + 1]) != 0; // opt-in on mesh scenes (raymond_hip.h)
			}
			Scene scene;
			if (what == "spheres") scene = reflective_spheres();
			else if (what.rfind("dragon", 0) == 0) scene = gold_dragon_standin(what.size() > 7 ? std::atoi(what.c_str() + 7) : 91);
			else if (what.rfind("project:", 0) == 0) scene = Project::load(what.substr(8)).build_scene(); // core/src/project.rs
			else throw Error(RMD_ERR_INVALID_ARGUMENT, "unknown scene " + what);
			TaskHandle handle = render_tiled(scene, st); // :152
			size_t progressed = 0;
			// progressive mode: every message is taken as it arrives (consume); else await() (:153)
			std::vector<Vector3> image = st.samples_per_iteration ? consume(handle, st, progressed) : handle.await();
			const size_t W = st.camera_settings.backbuffer_width, H = st.camera_settings.backbuffer_height;
			write_ppm(argv[7], tone_map(image), W, H); // :161-197
			if (!raw.empty()) {
				std::ofstream f(raw, std::ios::binary);
				f.write(reinterpret_cast<const char *>(image.data()), (std::streamsize)(image.size() * 24));
			}
			const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
			std::printf("Finished render.\nTotal render time: %.3fs\n%zu x %zu, %zu spp, %zu bounces: %.1f Msamples/s end to end\n", secs, W, H, st.sample_count,
			            st.bounce_limit, (double)W * H * st.sample_count / secs / 1e6);
			return 0;
		}
		std::fprintf(stderr, "usage: see the header of raymond_amd/host/cli.cpp\n");
		return 2;
	} catch (const std::exception &e) {
		std::fprintf(stderr, "raymond_cli: %s\n", e.what());
		return 1;
	}
}
