// Multi-GPU framebuffer reduce over RCCL (xGMI).  The reference has no counterpart: its only
// parallelism is the in-process worker pool (src/trace.rs:175-222).  Here each GPU renders a
// disjoint subset of the host tiles into a zero-initialised full-size f64 framebuffer and ONE
// ncclReduce(sum) to the root assembles the image; every pixel is non-zero on exactly one rank,
// so the sum is bit-identical to a single-GPU render.
// librccl is resolved with dlopen at first use so that loading libraymond_hip.so (and the
// single-GPU path) never depends on it.
#define RMD_WITH_HIP 1
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "internal.hpp"

static_assert(sizeof(ncclUniqueId) == RMD_COMM_ID_BYTES, "ncclUniqueId size");

struct rmd_comm {
	rmd_context *ctx = nullptr;
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1;
};

namespace {

// RCCL shares device buffers between the ranks of a node through HIP IPC.  On hosts whose driver offers dmabuf IPC only (this pool's: without
// it ncclCommInitRank fails with `hipIpcGetMemHandle: invalid argument`) the HSA runtime must see HSA_ENABLE_IPC_MODE_LEGACY=0 — and it reads
// its environment once, at the process's first HIP call.  The library does NOT touch the environment by itself (round 4 did, from a load-time
// constructor: that changed the HSA runtime's behaviour for single-GPU callers and for every other HIP user of the process, and raced with
// getenv in other threads).  A multi-GPU caller opts in with rmd_comm_prepare_process() before its first HIP call, or exports the variable;
// rmd_comm_create names the variable when RCCL's initialisation fails without it.
constexpr const char *kIpcVar = "HSA_ENABLE_IPC_MODE_LEGACY";

struct Rccl {
	void *handle = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	std::string error;
};

Rccl &rccl() {
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char *n : names) {
			r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
			if (r.handle) break;
		}
		if (!r.handle) {
			r.error = std::string("dlopen(librccl): ") + dlerror();
			return;
		}
		r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.handle, "ncclGetUniqueId");
		r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.handle, "ncclCommInitRank");
		r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
		r.Reduce = (decltype(r.Reduce))dlsym(r.handle, "ncclReduce");
		r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
		if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Reduce || !r.GetErrorString) r.error = "librccl lacks a required symbol";
	});
	return r;
}

rmd_status rccl_fail(rmd_context *ctx, const char *what, ncclResult_t e) {
	return rmd::fail(ctx, RMD_ERR_RCCL, std::string(what) + ": " + rccl().GetErrorString(e));
}

} // namespace

extern "C" {

rmd_status rmd_comm_prepare_process(void) {
	if (setenv(kIpcVar, "0", /*overwrite=*/0) != 0) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_comm_prepare_process: setenv failed");
	return RMD_OK;
}

rmd_status rmd_comm_unique_id(uint8_t id_out[RMD_COMM_ID_BYTES]) {
	if (!id_out) return rmd::fail(nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_comm_unique_id: null pointer");
	Rccl &r = rccl();
	if (!r.error.empty()) return rmd::fail(nullptr, RMD_ERR_RCCL, r.error);
	ncclUniqueId id;
	ncclResult_t e = r.GetUniqueId(&id);
	if (e != ncclSuccess) return rccl_fail(nullptr, "ncclGetUniqueId", e);
	std::memcpy(id_out, &id, RMD_COMM_ID_BYTES);
	return RMD_OK;
}

static rmd_status comm_create_impl(rmd_context *ctx, const uint8_t id[RMD_COMM_ID_BYTES], int32_t rank, int32_t world_size, rmd_comm **out) {
	if (!ctx || !id || !out || world_size < 1 || rank < 0 || rank >= world_size)
		return rmd::fail(ctx, RMD_ERR_INVALID_ARGUMENT, "rmd_comm_create: bad argument");
	Rccl &r = rccl();
	if (!r.error.empty()) return rmd::fail(ctx, RMD_ERR_RCCL, r.error);
	hipError_t he = hipSetDevice(ctx->device);
	if (he != hipSuccess) return rmd::fail(ctx, RMD_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
	rmd_comm *c = new (std::nothrow) rmd_comm();
	if (!c) return rmd::fail(ctx, RMD_ERR_OUT_OF_MEMORY, "rmd_comm_create: allocation failed");
	c->ctx = ctx, c->rank = rank, c->world = world_size;
	ncclUniqueId uid;
	std::memcpy(&uid, id, RMD_COMM_ID_BYTES);
	ncclResult_t e = r.CommInitRank(&c->comm, world_size, uid, rank);
	if (e != ncclSuccess) {
		delete c;
		if (world_size > 1 && std::getenv(kIpcVar) == nullptr)
			return rmd::fail(ctx, RMD_ERR_RCCL, std::string("ncclCommInitRank: ") + r.GetErrorString(e) +
			                                        " — HSA_ENABLE_IPC_MODE_LEGACY is not set in this process: on hosts whose driver offers dmabuf IPC only, every rank "
			                                        "needs HSA_ENABLE_IPC_MODE_LEGACY=0 before its first HIP call (export it, or call rmd_comm_prepare_process() before "
			                                        "rmd_context_create)");
		return rccl_fail(ctx, "ncclCommInitRank", e);
	}
	*out = c;
	return RMD_OK;
}
rmd_status rmd_comm_create(rmd_context *ctx, const uint8_t id[RMD_COMM_ID_BYTES], int32_t rank, int32_t world_size, rmd_comm **out) {
	if (out) *out = nullptr;
	return rmd::guarded(ctx, "rmd_comm_create", [&] { return comm_create_impl(ctx, id, rank, world_size, out); }); // (the loader's error texts are strings)
}

void rmd_comm_destroy(rmd_comm *comm) {
	if (!comm) return;
	if (comm->ctx) {
		(void)hipSetDevice(comm->ctx->device);
		(void)hipStreamSynchronize(comm->ctx->stream);
	}
	if (comm->comm) (void)rccl().CommDestroy(comm->comm);
	delete comm;
}

rmd_status rmd_reduce_framebuffer_async(rmd_comm *comm, double *accum_dev, size_t n_doubles, int32_t root) {
	if (!comm || !accum_dev || root < 0 || root >= comm->world) return rmd::fail(comm ? comm->ctx : nullptr, RMD_ERR_INVALID_ARGUMENT, "rmd_reduce_framebuffer: bad argument");
	rmd_context *ctx = comm->ctx;
	hipError_t he = hipSetDevice(ctx->device);
	if (he != hipSuccess) return rmd::fail(ctx, RMD_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(he));
	// on the context's stream: ordered behind the renders enqueued before it, in front of whatever the caller enqueues next
	ncclResult_t e = rccl().Reduce(accum_dev, accum_dev, n_doubles, ncclDouble, ncclSum, root, comm->comm, ctx->stream);
	if (e != ncclSuccess) return rccl_fail(ctx, "ncclReduce", e);
	return RMD_OK;
}

rmd_status rmd_reduce_framebuffer(rmd_comm *comm, double *accum_dev, size_t n_doubles, int32_t root) {
	if (rmd_status s = rmd_reduce_framebuffer_async(comm, accum_dev, n_doubles, root)) return s;
	rmd_context *ctx = comm->ctx;
	hipError_t he = hipStreamSynchronize(ctx->stream);
	if (he != hipSuccess) return rmd::fail(ctx, RMD_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(he));
	return rmd::check_fault(ctx); // the frames that were summed came from launches that may have been cut short
}

} // extern "C"
