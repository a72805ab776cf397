// Multi-GPU composition of one frame behind the C ABI (include/yafaray_c_api.h "multi-GPU"):
// pixel tiles are dealt to the ranks (yafaray_setShard), every rank renders its tiles into a full-frame film, and ONE
// ncclReduce(sum, root) of that [H][W][5] float film over RCCL / xGMI assembles the frame — the semantics of the reference's
// own film merge (sum colour, sum weight, normalise afterwards: src/common/imagefilm.cc:1467-1557).  A C or C++ host needs
// nothing but this library and librccl for it; no Python, no torch.
//
// RCCL is bound at run time (dlopen): a one-GPU host does not need the library, and inside a process that already holds a copy
// (PyTorch ships its own librccl.so) that copy is used instead of a second one.  Every entry point fails loudly
// (yafaray_commLastError) when RCCL is missing or a call fails; there is no host-staged fallback.
#include "../../include/yafaray_c_api.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace
{
struct Rccl
{
	void *lib = nullptr;
	std::string origin;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local std::string g_comm_error;

bool comm_fail(const std::string &msg) { g_comm_error = msg; return false; }

void load_rccl()
{
	// a copy already in the process first (RTLD_NOLOAD), then the system's
	struct Try { const char *name; int flags; };
	const char *env = std::getenv("YAFARAY_RCCL_LIB");
	const Try tries[] = {
		{env, RTLD_NOW | RTLD_LOCAL},
		{"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD}, {"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD},
		{"librccl.so.1", RTLD_NOW | RTLD_LOCAL}, {"librccl.so", RTLD_NOW | RTLD_LOCAL}, {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL},
	};
	for(const Try &t : tries)
	{
		if(!t.name || !*t.name) continue;
		void *h = dlopen(t.name, t.flags);
		if(!h) continue;
		Rccl r;
		r.lib = h;
		r.origin = std::string(t.name) + ((t.flags & RTLD_NOLOAD) ? " (already in the process)" : "");
		r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
		r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
		r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
		r.Reduce = (decltype(r.Reduce))dlsym(h, "ncclReduce");
		r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
		r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
		if(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Reduce && r.AllReduce) { g_rccl = r; return; }
		dlclose(h);
	}
}

bool have_rccl()
{
	std::call_once(g_rccl_once, load_rccl);
	if(!g_rccl.lib) return comm_fail("RCCL (librccl.so) could not be loaded: a multi-GPU film reduce is not possible on this host (YAFARAY_RCCL_LIB names another path)");
	return true;
}

std::string nccl_msg(const char *what, ncclResult_t r)
{
	std::string s = std::string(what) + " failed: ";
	s += g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "ncclResult";
	s += " (" + std::to_string((int)r) + ")";
	return s;
}
} // namespace

struct yafaray_comm
{
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1, device = 0;
	hipStream_t side = nullptr;      // exchanges between passes run here
};

extern "C" {

const char *yafaray_commLastError(void) { return g_comm_error.c_str(); }

const char *yafaray_commBackend(void) { return have_rccl() ? g_rccl.origin.c_str() : ""; }

yafaray_bool_t yafaray_commGetUniqueId(char id[YAFARAY_COMM_ID_BYTES])
{
	static_assert(YAFARAY_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is ncclUniqueId's payload");
	if(!id) return comm_fail("commGetUniqueId: null buffer");
	if(!have_rccl()) return 0;
	ncclUniqueId u;
	const ncclResult_t r = g_rccl.GetUniqueId(&u);
	if(r != ncclSuccess) return comm_fail(nccl_msg("ncclGetUniqueId", r));
	std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
	return 1;
}

yafaray_comm_t *yafaray_commCreate(const char id[YAFARAY_COMM_ID_BYTES], int rank, int world, int device)
{
	if(!id || world < 1 || rank < 0 || rank >= world) { comm_fail("commCreate: bad rank / world size"); return nullptr; }
	if(!have_rccl()) return nullptr;
	if(hipSetDevice(device) != hipSuccess) { comm_fail("commCreate: hipSetDevice(" + std::to_string(device) + ") failed"); return nullptr; }
	ncclUniqueId u;
	std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
	yafaray_comm *c = new yafaray_comm;
	c->rank = rank; c->world = world; c->device = device;
	const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);
	if(r != ncclSuccess) { comm_fail(nccl_msg("ncclCommInitRank", r)); delete c; return nullptr; }
	if(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) c->side = nullptr;
	return c;
}

void yafaray_commDestroy(yafaray_comm_t *c)
{
	if(!c) return;
	if(c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
	if(c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
	delete c;
}

int yafaray_commRank(const yafaray_comm_t *c) { return c ? c->rank : 0; }
int yafaray_commWorld(const yafaray_comm_t *c) { return c ? c->world : 1; }

// Interface-level: the frame's one collective.  In place on every rank; only `root` holds the sum afterwards.
yafaray_bool_t yafaray_reduceFilm(yafaray_comm_t *c, float *d_film, uint64_t n_floats, int root, void *stream)
{
	if(!c || !c->comm) return comm_fail("reduceFilm: no communicator");
	if(!d_film || root < 0 || root >= c->world) return comm_fail("reduceFilm: bad arguments");
	const ncclResult_t r = g_rccl.Reduce(d_film, d_film, (size_t)n_floats, ncclFloat, ncclSum, root, c->comm, (hipStream_t)stream);
	if(r != ncclSuccess) return comm_fail(nccl_msg("ncclReduce", r));
	return 1;
}

yafaray_bool_t yafaray_allReduce(yafaray_comm_t *c, float *d_values, uint64_t n_floats, void *stream)
{
	if(!c || !c->comm) return comm_fail("allReduce: no communicator");
	if(!d_values) return comm_fail("allReduce: null buffer");
	const ncclResult_t r = g_rccl.AllReduce(d_values, d_values, (size_t)n_floats, ncclFloat, ncclSum, c->comm, (hipStream_t)stream);
	if(r != ncclSuccess) return comm_fail(nccl_msg("ncclAllReduce", r));
	return 1;
}

// the yafaray_plane_exchange_t of a communicator: what yafaray_setComm attaches (plane exchange between adaptive passes,
// light-counter exchange of the serial replay).  The caller has synchronised the device (yafgpu_device.hip); the sum is complete on return.
int yafaray_commExchange(void *user, float *d_values, uint64_t n_floats)
{
	yafaray_comm *c = (yafaray_comm *)user;
	if(!c || c->world <= 1) return 0;
	if(!yafaray_allReduce(c, d_values, n_floats, c->side)) { std::fprintf(stderr, "[yafaray] %s\n", g_comm_error.c_str()); return 1; }
	return hipStreamSynchronize(c->side) == hipSuccess ? 0 : 1;
}

} // extern "C"
