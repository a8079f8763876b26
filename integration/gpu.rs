//! `src/gpu.rs` for Nyrox/raymond — binds `libraymond_hip.so` (include/raymond_hip.h) and replaces the per-tile body of
//! `render_tiled` (src/trace.rs:197-205) with one call per batch of tiles.
//!
//! SOURCE ONLY: no Rust toolchain exists in the image this repository is built in, so this file has never been compiled.
//! The same call sequence is compiled, run and tested through the C++ mirror (`raymond_amd/host/raymond.cpp`,
//! `tests/test_host_cpp.py`) and the Python mirror (`raymond_amd/render.py`).  To adopt it:
//!   * add `mod gpu;` to `src/lib.rs` and `println!("cargo:rustc-link-search=native=<repo>/raymond_amd/csrc")` to `build.rs`;
//!   * give `core::geometry::acc_grid::Cell` a `pub fn index(&self) -> usize { self.0 }` (its field is private);
//!   * call `gpu::render_tiled_gpu(scene, settings, seed)` where `render_tiled(scene, settings)` is called today
//!     (cli_old/src/main.rs:152) — it returns the same `TaskHandle`, so `await`/`poll`/`async_await` are unchanged.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_void};
use std::ptr;
use std::sync::atomic::{AtomicUsize, Ordering};
use std::sync::{mpsc, Arc};
use std::thread;

use crossbeam::queue::MsQueue;

use core::geometry::AccGrid;
use core::scene::{Geometry, Scene};
use core::tile::Tile;
use core::{Material, Vector3};

use super::trace::{Message, Settings, TaskHandle};

// ---------------------------------------------------------------- the C ABI (include/raymond_hip.h)
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rmd_material { pub kind: u32, _pad: u32, pub color: [f64; 3], pub roughness: f64, pub emission_aux: [f64; 5] }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rmd_object { pub geometry_kind: u32, pub grid_index: u32, pub origin: [f64; 3], pub normal: [f64; 3], pub radius: f64, pub material: rmd_material }
#[repr(C)]
pub struct rmd_grid_desc {
    pub bbox_min: [f64; 3], pub bbox_max: [f64; 3], pub resolution: [u32; 3], _pad: u32, pub cell_size: [f64; 3],
    pub cells: *const u32, pub n_cells: u64, pub mapping_table: *const u32, pub n_mapping: u64,
    pub tri_pos: *const f64, pub tri_nrm: *const f64, pub n_tris: u64,
    pub built: *const c_void, // NULL, or the rmd_grid_build the pointers belong to (rmd_grid_build_describe sets it: device tables derived once per build)
}
#[repr(C)]
pub struct rmd_camera { pub backbuffer_width: u32, pub backbuffer_height: u32, pub fov_vert: f64, pub position: [f64; 3], pub focal_length: f64, pub aperture_radius: f64 }
#[repr(C)]
pub struct rmd_settings { pub bounce_limit: u32, pub sample_begin: u32, pub sample_count: u32, pub flags: u32 /* 0 = the reference-identical mode; RMD_RENDER_DOF = 1, RMD_RENDER_TRACE_BLACK_PATHS = 2, RMD_RENDER_END_BLACK_PATHS = 4 */, pub seed: u64 }
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rmd_tile_rect { pub left: u32, pub top: u32, pub width: u32, pub height: u32 }
pub enum rmd_context {}
pub enum rmd_scene {}

#[link(name = "raymond_hip")]
/// include/raymond_hip.h: RMD_ABI_VERSION this module's struct definitions were written against
const RMD_ABI_VERSION: u32 = 5;

extern "C" {
    fn rmd_abi_version() -> u32;
    fn rmd_context_create(device_ordinal: i32, out: *mut *mut rmd_context) -> i32;
    fn rmd_context_destroy(ctx: *mut rmd_context);
    fn rmd_last_error(ctx: *const rmd_context) -> *const c_char;
    fn rmd_scene_create(ctx: *mut rmd_context, objects: *const rmd_object, n_objects: u32, grids: *const rmd_grid_desc, n_grids: u32, out: *mut *mut rmd_scene) -> i32;
    fn rmd_scene_destroy(scene: *mut rmd_scene);
    fn rmd_framebuffer_alloc(ctx: *mut rmd_context, width: u32, height: u32, out: *mut *mut f64) -> i32;
    fn rmd_framebuffer_free(ctx: *mut rmd_context, dev: *mut f64) -> i32;
    // tile rectangles <-> a packed buffer in Tile.data layout (tile sums stay resident on the GPU; only what a message carries is moved)
    fn rmd_framebuffer_upload_tiles(ctx: *mut rmd_context, host_packed: *const f64, dev: *mut f64, width: u32, height: u32, rects: *const rmd_tile_rect, n_rects: u32) -> i32;
    fn rmd_framebuffer_download_tiles(ctx: *mut rmd_context, dev: *const f64, width: u32, height: u32, rects: *const rmd_tile_rect, n_rects: u32, host_packed: *mut f64) -> i32;
    fn rmd_render_tiles(ctx: *mut rmd_context, scene: *const rmd_scene, camera: *const rmd_camera, settings: *const rmd_settings,
                        tiles: *const rmd_tile_rect, n_tiles: u32, accum_dev: *mut f64) -> i32;
}

fn check(ctx: *const rmd_context, status: i32) {
    if status != 0 {
        let text = unsafe { std::ffi::CStr::from_ptr(rmd_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("raymond_hip: status {}: {}", status, text); // the reference's failure mode is a panic
    }
}

// ---------------------------------------------------------------- flattening
fn v3(v: Vector3) -> [f64; 3] { [v.x, v.y, v.z] }

fn material(m: &Material) -> rmd_material {
    match *m {
        Material::Diffuse(c, r) => rmd_material { kind: 0, _pad: 0, color: v3(c), roughness: r, emission_aux: [0.0; 5] },
        Material::Metal(c, r) => rmd_material { kind: 1, _pad: 0, color: v3(c), roughness: r, emission_aux: [0.0; 5] },
        Material::Emission(e, v2, f1, f2) => rmd_material { kind: 2, _pad: 0, color: v3(e), roughness: 0.0, emission_aux: [v2.x, v2.y, v2.z, f1, f2] },
    }
}

/// Owns the compact copies of every distinct `Arc<AccGrid>` of the scene; the descriptors point into them.
struct FlatGrids { cells: Vec<Vec<u32>>, maps: Vec<Vec<u32>>, pos: Vec<Vec<f64>>, nrm: Vec<Vec<f64>>, descs: Vec<rmd_grid_desc> }

fn flatten(scene: &Scene) -> (Vec<rmd_object>, FlatGrids) {
    let mut grids: Vec<Arc<AccGrid>> = Vec::new();
    let mut objects = Vec::with_capacity(scene.objects.len());
    for o in &scene.objects { // object order is significant: Scene::intersect keeps the first object on ties (core/src/scene.rs:61)
        let zero = [0.0; 3];
        let (kind, grid_index, origin, normal, radius) = match &o.geometry {
            Geometry::Plane(p) => (0, 0, v3(p.origin), v3(p.normal), 0.0),
            Geometry::Sphere(s) => (1, 0, v3(s.origin), zero, s.radius),
            Geometry::Grid(g) => {
                let idx = grids.iter().position(|h| Arc::ptr_eq(h, g)).unwrap_or_else(|| { grids.push(g.clone()); grids.len() - 1 });
                (2, idx as u32, zero, zero, 0.0)
            }
        };
        objects.push(rmd_object { geometry_kind: kind, grid_index, origin, normal, radius, material: material(&o.material) });
    }
    let mut f = FlatGrids { cells: vec![], maps: vec![], pos: vec![], nrm: vec![], descs: vec![] };
    for g in &grids {
        f.cells.push(g.cells.iter().map(|c| c.index() as u32).collect());
        f.maps.push(g.mapping_table.iter().map(|&i| i as u32).collect());
        let mut pos = Vec::with_capacity(g.mesh.triangles.len() * 9);
        let mut nrm = Vec::with_capacity(g.mesh.triangles.len() * 9);
        for t in &g.mesh.triangles {
            for v in &[t.0, t.1, t.2] { pos.extend_from_slice(&v3(v.position)); }
            for v in &[t.0, t.1, t.2] { nrm.extend_from_slice(&v3(v.normal)); }
        }
        f.pos.push(pos);
        f.nrm.push(nrm);
    }
    for (i, g) in grids.iter().enumerate() {
        f.descs.push(rmd_grid_desc {
            bbox_min: v3(g.mesh.bounding_box.min), bbox_max: v3(g.mesh.bounding_box.max),
            resolution: [g.resolution.x as u32, g.resolution.y as u32, g.resolution.z as u32], _pad: 0, cell_size: v3(g.cell_size),
            cells: f.cells[i].as_ptr(), n_cells: f.cells[i].len() as u64, mapping_table: f.maps[i].as_ptr(), n_mapping: f.maps[i].len() as u64,
            tri_pos: f.pos[i].as_ptr(), tri_nrm: f.nrm[i].as_ptr(), n_tris: g.mesh.triangles.len() as u64,
            built: ptr::null(), // these tables are the reference's own AccGrid, converted: nothing to reuse
        });
    }
    (objects, f)
}

// ---------------------------------------------------------------- render_tiled with GPU workers
/// `render_tiled` (src/trace.rs:137-230) with `settings.worker_count` GPUs instead of CPU threads.  Tile generation, the
/// queue, `Message` and `TaskHandle` are the reference's; a worker pops a batch of tiles that stand at the same sample
/// count and renders one pass for all of them in a single kernel launch.
pub fn render_tiled_gpu(scene: Scene, settings: Settings, seed: u64) -> TaskHandle {
    let queue = Arc::new(MsQueue::new());
    let (sender, receiver) = mpsc::channel();
    super::trace::push_tiles(&queue, &settings); // the 'gen_tiles loop of :142-173, moved into a function

    let thread_count = Arc::new(AtomicUsize::new(settings.worker_count));
    // tiles a worker has popped and not yet finished or re-queued: a worker may only leave when the queue is empty AND this is 0
    // (with one GPU call per batch the queue is momentarily empty between progressive passes while other workers hold the tiles)
    let in_flight = Arc::new(AtomicUsize::new(0));
    let (tw, th) = settings.tile_size;
    let cs = &settings.camera_settings;
    let n_tiles = ((cs.backbuffer_width + tw - 1) / tw) * ((cs.backbuffer_height + th - 1) / th);
    // a quarter of a GPU's share per pop, so that every GPU gets work and the tail stays short (1080p: 2040 tiles, 8 GPUs -> 64)
    let gpu_workers = settings.worker_count;
    // one GPU: every tile in one launch per pass; several: a quarter of a GPU's share per pop
    let batch_size = if gpu_workers == 1 { n_tiles } else { ((n_tiles + gpu_workers * 4 - 1) / (gpu_workers * 4)).max(1) };
    for gpu in 0..settings.worker_count {
        let (queue, sender, thread_count, in_flight) = (queue.clone(), sender.clone(), thread_count.clone(), in_flight.clone());
        let (scene, settings) = (scene.clone(), settings.clone()); // :182-185
        thread::spawn(move || unsafe {
            // (the structs above have grown from ABI version to version: a library of another version is refused before the first other call)
            assert_eq!(rmd_abi_version(), RMD_ABI_VERSION, "libraymond_hip.so is of another ABI version than this module");
            let mut ctx = ptr::null_mut();
            check(ptr::null(), rmd_context_create(gpu as i32, &mut ctx));
            let (objects, grids) = flatten(&scene);
            let mut dev_scene = ptr::null_mut();
            check(ctx, rmd_scene_create(ctx, objects.as_ptr(), objects.len() as u32, grids.descs.as_ptr(), grids.descs.len() as u32, &mut dev_scene));
            let cam = &settings.camera_settings;
            let (w, h) = (cam.backbuffer_width, cam.backbuffer_height);
            let camera = rmd_camera { backbuffer_width: w as u32, backbuffer_height: h as u32, fov_vert: cam.fov_vert,
                                      position: v3(cam.transform.position), focal_length: cam.focal_length, aperture_radius: cam.aperture_radius };
            let mut fb = ptr::null_mut();
            check(ctx, rmd_framebuffer_alloc(ctx, w as u32, h as u32, &mut fb));
            // The tiles' running sums stay RESIDENT in `fb` (zeroed by rmd_framebuffer_alloc: a fresh tile's sums) between passes; what crosses the bus
            // is what a message carries.  With ONE GPU worker every tile's sums are always in this framebuffer; with several, a tile that goes back to
            // the shared queue takes its sums along in tile.data and the worker that pops it next uploads them: below, every popped tile with
            // `begin != 0` is uploaded when `gpu_workers > 1` (no per-tile table of who holds what: the worker cannot know who rendered the tile last).
            let mut packed: Vec<f64> = Vec::new();
            // samples per pass: all of them at once, or `samples_per_iteration` when progress messages are wanted (:217)
            let pass = if settings.samples_per_iteration != 0 { settings.samples_per_iteration } else { settings.sample_count };
            loop {
                // a batch of tiles at the same sample count (was: one tile, `queue.try_pop()`, :189)
                let mut batch: Vec<Tile> = Vec::new();
                while batch.len() < batch_size {
                    in_flight.fetch_add(1, Ordering::AcqRel); // counted before the pop so that the sum never reads 0 while a tile is in hand
                    match queue.try_pop() {
                        Some(t) => { if batch.first().map_or(true, |b: &Tile| b.sample_count == t.sample_count) { batch.push(t) } else { queue.push(t); in_flight.fetch_sub(1, Ordering::AcqRel); break } }
                        None => { in_flight.fetch_sub(1, Ordering::AcqRel); break }
                    }
                }
                if batch.is_empty() {
                    if in_flight.load(Ordering::Acquire) != 0 { thread::yield_now(); continue; } // other workers will re-queue their tiles
                    thread_count.fetch_sub(1, Ordering::Relaxed); // :191-193
                    break;
                }
                let begin = batch[0].sample_count;
                let n = pass.min(settings.sample_count - begin);
                let rects: Vec<rmd_tile_rect> = batch.iter().map(|t| rmd_tile_rect { left: t.left as u32, top: t.top as u32, width: t.width as u32, height: t.height as u32 }).collect();
                // several GPU workers: tiles whose earlier passes another GPU rendered arrive with their sums in tile.data (Tile.data layout IS the
                // packed layout of rmd_framebuffer_upload_tiles: row-major within the tile, one tile after the other)
                if gpu_workers > 1 && begin != 0 {
                    packed.clear();
                    for tile in &batch { for v in &tile.data { packed.extend_from_slice(&[v.x, v.y, v.z]); } }
                    check(ctx, rmd_framebuffer_upload_tiles(ctx, packed.as_ptr(), fb, w as u32, h as u32, rects.as_ptr(), rects.len() as u32));
                }
                // flags: 0 — the drop-in mode.  The reference's loop calls the pinhole generate_primary_ray whatever cam.aperture_radius holds (:199);
                // RMD_RENDER_DOF (1) would opt into generate_primary_ray_with_dof, which the reference defines but never calls.  With 0 every sample
                // is the reference's, NaN for NaN: paths whose throughput is exactly zero are ended only in scenes without a mesh whose parameters
                // are all finite and regular, where that is provably exact; RMD_RENDER_END_BLACK_PATHS (4) would end them on mesh scenes too
                // (1.6x faster; a sample the reference makes 0 x NaN = NaN then comes out 0 — raymond_hip.h)
                let st = rmd_settings { bounce_limit: settings.bounce_limit as u32, sample_begin: begin as u32, sample_count: n as u32, flags: 0, seed };
                // the kernel adds samples begin..begin+n to the resident sums one by one (`+=` of :203, in sample order: progressive passes give
                // the same bits as one pass).  RMD_ERR_DEVICE_FAULT (8) here means the launch was cut short: check() panics, as the reference would
                check(ctx, rmd_render_tiles(ctx, dev_scene, &camera, &st, rects.as_ptr(), rects.len() as u32, fb)); // replaces :197-205
                // what has to come to the host: finished tiles (:211-212), progress snapshots (:217-219), tiles that may change GPU
                let finished = begin + n == settings.sample_count;
                let snapshot = !finished && settings.samples_per_iteration != 0;
                if finished || snapshot || gpu_workers > 1 {
                    packed.resize(rects.iter().map(|r| (r.width * r.height * 3) as usize).sum(), 0.0);
                    // (the C++ mirror, raymond_amd/host/raymond.cpp, uses rmd_framebuffer_download_tiles_async into page-locked memory
                    // (rmd_host_alloc) and sends the messages of batch k while batch k + 1 renders: rmd_context_wait_transfers)
                    check(ctx, rmd_framebuffer_download_tiles(ctx, fb, w as u32, h as u32, rects.as_ptr(), rects.len() as u32, packed.as_mut_ptr()));
                }
                let held = batch.len();
                let mut at = 0usize;
                for mut tile in batch {
                    let len = tile.width * tile.height;
                    if finished || snapshot || gpu_workers > 1 {
                        tile.data.clear();
                        tile.data.extend(packed[at * 3..(at + len) * 3].chunks_exact(3).map(|c| Vector3::new(c[0], c[1], c[2])));
                    }
                    at += len;
                    tile.sample_count += n; // :207
                    if tile.sample_count == settings.sample_count {
                        sender.send(Message::TileFinished(tile)).unwrap(); // :211-212
                    } else {
                        queue.push(tile.clone()); // :214 (one GPU: the clone's data is not read again — the sums are in `fb`)
                        if settings.samples_per_iteration != 0 { sender.send(Message::TileProgressed(tile)).unwrap(); } // :217-219
                    }
                }
                in_flight.fetch_sub(held, Ordering::AcqRel); // after the re-queue: the queue and this counter are never both empty mid-render
            }
            rmd_framebuffer_free(ctx, fb);
            rmd_scene_destroy(dev_scene);
            rmd_context_destroy(ctx);
        });
    }
    TaskHandle::new(receiver, settings, thread_count) // the struct literal of :224-229 (its fields are private to trace.rs)
}
