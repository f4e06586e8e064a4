// abi_comm_ring.h -- the C ABI entry points around the chain: RCCL communicator + all-gather of the result table (8e), the
// id-file bootstrap, the pinned ingest ring (8f-3).  Included by gsmcal.hip inside its extern "C" block.
#pragma once
// ---- multi-GPU: RCCL all-gather of the result table ------------------------------------------------------------
// librccl.so is loaded on first use, so single-GPU users of libgsmcal.so do not depend on it.
struct RcclApi {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi* rccl_api() {
    static RcclApi api;
    if (api.h) return &api;
    // RCCL must belong to the HIP runtime this process runs on: a process whose runtime is the copy bundled with a
    // PyTorch-ROCm wheel and whose RCCL is the system one works until exit and then aborts in the allocator (double free).
    // So: a librccl that is mapped already; else the one lying beside the loaded libamdhip64; else the loader's choice.
    // (RTLD_NODELETE throughout: RCCL registers exit-time clean-up of its own; a process that unloads the library before that
    // runs -- a Python interpreter tearing down its ctypes handles in no particular order -- ends in the allocator with
    // "double free or corruption" after all work is done and checked)
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD | RTLD_NODELETE);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD | RTLD_NODELETE);
    if (!h) {
        Dl_info di;
        if (dladdr((void*)&hipGetDeviceCount, &di) && di.dli_fname) {
            std::string dir(di.dli_fname);
            const size_t cut = dir.rfind('/');
            if (cut != std::string::npos) {
                dir.resize(cut + 1);
                h = dlopen((dir + "librccl.so").c_str(), RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
                if (!h) h = dlopen((dir + "librccl.so.1").c_str(), RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
            }
        }
    }
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NODELETE);
    if (!h) return nullptr;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.AllGather = (decltype(api.AllGather))dlsym(h, "ncclAllGather");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) return nullptr;
    api.h = h;
    return &api;
}
struct gsmcal_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
};
static_assert(GSMCAL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

int gsmcal_comm_get_unique_id(void* id_out) {
    if (!id_out) return GSMCAL_E_ARG;
    RcclApi* a = rccl_api();
    if (!a) return GSMCAL_E_UNSUPPORTED;
    ncclUniqueId id;
    if (a->GetUniqueId(&id) != ncclSuccess) return GSMCAL_E_HIP;
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int gsmcal_comm_init_rank(gsmcal_ctx* c, const void* idp, int world, int rank, gsmcal_comm** out) {
    if (!c || !idp || !out || world < 1 || rank < 0 || rank >= world) return GSMCAL_E_ARG;
    *out = nullptr;
    RcclApi* a = rccl_api();
    if (!a) { c->err = "librccl.so could not be loaded"; return GSMCAL_E_UNSUPPORTED; }
    ENTER(c);
    ncclUniqueId id;
    memcpy(&id, idp, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = a->CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) {
        c->err = std::string("ncclCommInitRank: ") + (a->GetErrorString ? a->GetErrorString(r) : "failed");
        return GSMCAL_E_HIP;
    }
    gsmcal_comm* g = new gsmcal_comm();
    g->comm = comm; g->world = world; g->rank = rank;
    *out = g;
    return 0;
}

// ---- id-file bootstrap: [8 B magic | 8 B nonce | 128 B id], run-specific (see include/gsmcal.h) ----
static const unsigned long long GSMCAL_ID_MAGIC = 0x3144494c41434d47ull;   // "GMCALID1"

int gsmcal_comm_id_file_remove(const char* path) {
    if (!path) return GSMCAL_E_ARG;
    (void)unlink(path);
    (void)unlink((std::string(path) + ".tmp").c_str());
    return 0;
}

// age_test: also reject a record older than the stale window (GSMCAL_COMM_STALE_S, 120 s).  Always on for nonce 0; on as well
// for a nonce that was only DERIVED from the environment (default_launch_nonce): plain torchrun gives every launch the same
// MASTER_ADDR:MASTER_PORT, so a derived nonce may repeat across launches and must not switch the age test off.
static int id_file_exchange(const char* path, unsigned long long nonce, bool age_test, int world, int rank, void* id_inout, double timeout_s) {
    if (!path || !id_inout || world < 1 || rank < 0 || rank >= world) return GSMCAL_E_ARG;
    const size_t rec = 16 + GSMCAL_COMM_ID_BYTES;
    unsigned char buf[16 + GSMCAL_COMM_ID_BYTES];
    if (rank == 0) {
        (void)gsmcal_comm_id_file_remove(path);                           // whatever an earlier (crashed) bootstrap left behind
        memcpy(buf, &GSMCAL_ID_MAGIC, 8);
        memcpy(buf + 8, &nonce, 8);
        memcpy(buf + 16, id_inout, GSMCAL_COMM_ID_BYTES);
        const std::string tmp = std::string(path) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(buf, 1, rec, f) != rec) { if (f) fclose(f); return GSMCAL_E_ARG; }
        if (fclose(f) != 0) return GSMCAL_E_ARG;
        if (rename(tmp.c_str(), path) != 0) return GSMCAL_E_ARG;          // atomic: readers never see half a record
        return 0;
    }
    double stale_s = 120.0;
    if (const char* e = getenv("GSMCAL_COMM_STALE_S")) { const double v = atof(e); if (v > 0.0) stale_s = v; }
    const long tries = (long)(timeout_s > 0.0 ? timeout_s * 100.0 : 6000.0);
    for (long t = 0; t < tries; ++t) {
        FILE* f = fopen(path, "rb");
        if (f) {
            const size_t got = fread(buf, 1, rec, f);
            const bool more = got == rec && fgetc(f) != EOF;
            struct stat sb;
            const bool have_sb = fstat(fileno(f), &sb) == 0;
            fclose(f);
            unsigned long long magic = 0, fn = 0;
            memcpy(&magic, buf, 8);
            memcpy(&fn, buf + 8, 8);
            bool ok = got == rec && !more && magic == GSMCAL_ID_MAGIC && fn == nonce;
            // no caller-chosen nonce to tell runs apart: a record older than the stale window belongs to a bootstrap that died
            if (ok && (nonce == 0 || age_test)) ok = have_sb && difftime(time(nullptr), sb.st_mtime) <= stale_s;
            if (ok) { memcpy(id_inout, buf + 16, GSMCAL_COMM_ID_BYTES); return 0; }
        }
        usleep(10000);
    }
    return GSMCAL_E_ARG;
}

int gsmcal_comm_id_file_exchange(const char* path, unsigned long long nonce, int world, int rank, void* id_inout, double timeout_s) {
    return id_file_exchange(path, nonce, nonce == 0, world, rank, id_inout, timeout_s);
}

int gsmcal_comm_id_file_exchange_aged(const char* path, unsigned long long nonce, int world, int rank, void* id_inout, double timeout_s) {
    return id_file_exchange(path, nonce, true, world, rank, id_inout, timeout_s);
}

static int comm_init_file(gsmcal_ctx* c, const char* path, unsigned long long nonce, bool age_test, int world, int rank, gsmcal_comm** out) {
    if (!c || !path || !out || world < 1 || rank < 0 || rank >= world) return GSMCAL_E_ARG;
    unsigned char id[GSMCAL_COMM_ID_BYTES];
    if (rank == 0) RET_IF(gsmcal_comm_get_unique_id(id));
    if (id_file_exchange(path, nonce, age_test, world, rank, id, 60.0) != 0) {
        c->err = rank == 0 ? "cannot publish the id file" : "timed out waiting for rank 0's id file (this run's nonce)";
        return GSMCAL_E_ARG;
    }
    const int rc = gsmcal_comm_init_rank(c, id, world, rank, out);
    if (rank == 0) (void)gsmcal_comm_id_file_remove(path);              // every rank has joined (or the bootstrap failed): the id is spent
    return rc;
}

int gsmcal_comm_init_file_nonce(gsmcal_ctx* c, const char* path, unsigned long long nonce, int world, int rank, gsmcal_comm** out) {
    return comm_init_file(c, path, nonce, nonce == 0, world, rank, out);
}

// The nonce gsmcal_comm_init_file uses when the caller names none: GSMCAL_COMM_NONCE if set, else a hash of what identifies
// this LAUNCH to every one of its ranks -- the launcher's run id (TORCHELASTIC_RUN_ID, unless it is torchrun's literal default
// "none") with its restart count, a batch scheduler's job id, and the rendezvous address (MASTER_ADDR:MASTER_PORT).  0 when the
// environment offers none of these.  *strong = the caller chose it (GSMCAL_COMM_NONCE): only then may readers skip the age
// test.  A derived nonce can repeat -- plain `torchrun` has RUN_ID "none" and the static 127.0.0.1:29500 in every launch -- so
// records carrying it are still held to the stale window (GSMCAL_COMM_STALE_S): an id file a crashed bootstrap left behind is
// rejected by the nonce when the launcher tells launches apart and by its age when it does not.
static unsigned long long default_launch_nonce(bool* strong = nullptr) {
    if (strong) *strong = false;
    if (const char* e = getenv("GSMCAL_COMM_NONCE")) { if (strong) *strong = true; return strtoull(e, nullptr, 0); }
    std::string id;
    for (const char* name : {"TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "SLURM_JOB_ID", "SLURM_STEP_ID", "PBS_JOBID", "LSB_JOBID"})
        if (const char* v = getenv(name)) {
            if (!*v) continue;
            if (!strcmp(name, "TORCHELASTIC_RUN_ID") && !strcmp(v, "none")) continue;       // torch.distributed.run's default: identifies nothing
            if (!strcmp(name, "TORCHELASTIC_RESTART_COUNT") && !strcmp(v, "0") && id.empty()) continue;   // (a first attempt without a run id says nothing either)
            id += name; id += '='; id += v; id += ';';
        }
    const char* ma = getenv("MASTER_ADDR");
    const char* mp = getenv("MASTER_PORT");
    if (ma && mp && *ma && *mp) { id += ma; id += ':'; id += mp; }
    if (id.empty()) return 0;
    unsigned long long h = 0xcbf29ce484222325ull;           // FNV-1a, 64 bit
    for (unsigned char ch : id) { h ^= ch; h *= 0x100000001b3ull; }
    return h ? h : 1;
}

unsigned long long gsmcal_comm_default_nonce(void) { return default_launch_nonce(); }

int gsmcal_comm_init_file(gsmcal_ctx* c, const char* path, int world, int rank, gsmcal_comm** out) {
    bool strong = false;
    const unsigned long long nonce = default_launch_nonce(&strong);
    return comm_init_file(c, path, nonce, !strong, world, rank, out);
}

void gsmcal_comm_destroy(gsmcal_comm* g) {
    if (!g) return;
    RcclApi* a = rccl_api();
    if (a && g->comm) (void)a->CommDestroy(g->comm);
    delete g;
}

int gsmcal_allgather_table(gsmcal_ctx* c, gsmcal_comm* g, const double* d_local, int rows_per_rank, int cols, double* d_all) {
    if (!c || !g || !d_local || !d_all || rows_per_rank < 1 || cols < 1) return GSMCAL_E_ARG;
    RcclApi* a = rccl_api();
    if (!a) return GSMCAL_E_UNSUPPORTED;
    HIPCHK(c, hipSetDevice(c->device));
    // behind a pipelined batch call the collective rides on the stream that call's table is written on (its last stage), and the
    // gathered table is complete where the call's own outputs are: `depth` calls later in the context's stream order, or gsmcal_sync
    const hipStream_t st = pipe_out_stream(c);
    if (st != c->stream && c->ag_chain_n > 0) HIPCHK(c, hipStreamWaitEvent(st, c->ag_chain[(c->ag_chain_n - 1) & 1], 0));   // behind the previous call's collective
    const ncclResult_t r = a->AllGather(d_local, d_all, (size_t)rows_per_rank * cols, ncclDouble, g->comm, st);
    if (r != ncclSuccess) {
        c->err = std::string("ncclAllGather: ") + (a->GetErrorString ? a->GetErrorString(r) : "failed");
        return GSMCAL_E_HIP;
    }
    if (st != c->stream) {
        hipEvent_t& ev = c->ag_chain[c->ag_chain_n & 1];
        if (!ev) HIPCHK(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventReleaseToDevice));
        HIPCHK(c, hipEventRecord(ev, st));
        ++c->ag_chain_n;
        HIPCHK(c, hipEventRecord(c->pipe_done[c->pipe_last_slot], st));   // the call's completion now includes its collective
    }
    return 0;
}

// The same collective OFF the chain's critical path: RCCL runs on a side stream of the context, ordered
// behind an event recorded on the context's stream now; the context's stream itself does not wait, so the next batch's
// kernels start at once and the gather of batch i travels under the kernels of batch i+1.
int gsmcal_allgather_table_async(gsmcal_ctx* c, gsmcal_comm* g, const double* d_local, int rows_per_rank, int cols, double* d_all, int slot) {
    if (!c || !g || !d_local || !d_all || rows_per_rank < 1 || cols < 1 || slot < 0 || slot >= gsmcal_ctx::AG_SLOTS) return GSMCAL_E_ARG;
    RcclApi* a = rccl_api();
    if (!a) return GSMCAL_E_UNSUPPORTED;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->ag_stream) {
        int lo = 0, hi = 0;                                     // lowest priority: the collective never delays the chain's kernels
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&c->ag_stream, hipStreamNonBlocking, lo) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(c, hipStreamCreateWithFlags(&c->ag_stream, hipStreamNonBlocking));
        }
    }
    if (!c->ag_ready[slot]) {
        // device-scope release: the table only has to be visible to the collective's kernel on this device; a default event
        // flushes to system scope at every record (see get_event())
        const unsigned fl = hipEventDisableTiming | hipEventReleaseToDevice;     // (no other combination changed the event's cost: NOTES_r04)
        if (hipEventCreateWithFlags(&c->ag_ready[slot], fl) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(c, hipEventCreateWithFlags(&c->ag_ready[slot], hipEventDisableTiming));
        }
    }
    if (!c->ag_done[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->ag_done[slot], hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ag_ready[slot], pipe_out_stream(c)));   // (behind a pipelined batch call: the stream its table is written on)
    HIPCHK(c, hipStreamWaitEvent(c->ag_stream, c->ag_ready[slot], 0));
    const ncclResult_t r = a->AllGather(d_local, d_all, (size_t)rows_per_rank * cols, ncclDouble, g->comm, c->ag_stream);
    if (r != ncclSuccess) {
        c->err = std::string("ncclAllGather: ") + (a->GetErrorString ? a->GetErrorString(r) : "failed");
        return GSMCAL_E_HIP;
    }
    HIPCHK(c, hipEventRecord(c->ag_done[slot], c->ag_stream));
    c->ag_posted[slot] = true;
    return 0;
}

int gsmcal_allgather_wait(gsmcal_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= gsmcal_ctx::AG_SLOTS) return GSMCAL_E_ARG;
    if (c->ag_posted[slot]) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ag_done[slot], 0));
    return 0;
}

int gsmcal_allgather_sync(gsmcal_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= gsmcal_ctx::AG_SLOTS) return GSMCAL_E_ARG;
    if (c->ag_posted[slot]) HIPCHK(c, hipEventSynchronize(c->ag_done[slot]));
    return 0;
}

// ---- ingest ring ---------------------------------------------------------------------------------------------------
struct gsmcal_ring {
    gsmcal_ctx* c = nullptr;
    size_t bytes = 0;
    int n = 0;
    hipStream_t copy = nullptr;
    std::vector<void*> host, dev;
    std::vector<hipEvent_t> copied, consumed;      // H2D of the slot done / consumer kernels of the slot done
    std::vector<char> has_consumed;
};

int gsmcal_ring_create(gsmcal_ctx* c, size_t batch_bytes, int slots, gsmcal_ring** out) {
    if (!c || !out || batch_bytes < 1 || slots < 2 || slots > 16) return GSMCAL_E_ARG;
    *out = nullptr;
    ENTER(c);
    gsmcal_ring* r = new gsmcal_ring();
    r->c = c; r->bytes = batch_bytes; r->n = slots;
    r->host.assign(slots, nullptr); r->dev.assign(slots, nullptr);
    r->copied.assign(slots, nullptr); r->consumed.assign(slots, nullptr); r->has_consumed.assign(slots, 0);
    bool ok = hipStreamCreateWithFlags(&r->copy, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < slots; ++i) {
        ok = hipHostMalloc(&r->host[i], batch_bytes, hipHostMallocDefault) == hipSuccess &&
             hipMalloc(&r->dev[i], batch_bytes) == hipSuccess &&
             hipEventCreateWithFlags(&r->copied[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&r->consumed[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { c->err = "ring allocation failed"; gsmcal_ring_destroy(r); return GSMCAL_E_HIP; }
    *out = r;
    return 0;
}

void gsmcal_ring_destroy(gsmcal_ring* r) {
    if (!r) return;
    (void)hipSetDevice(r->c->device);
    if (r->copy) (void)hipStreamSynchronize(r->copy);
    (void)hipStreamSynchronize(r->c->stream);
    for (int i = 0; i < r->n; ++i) {
        if (r->host[i]) (void)hipHostFree(r->host[i]);
        if (r->dev[i]) (void)hipFree(r->dev[i]);
        if (r->copied[i]) (void)hipEventDestroy(r->copied[i]);
        if (r->consumed[i]) (void)hipEventDestroy(r->consumed[i]);
    }
    if (r->copy) (void)hipStreamDestroy(r->copy);
    delete r;
}

void* gsmcal_ring_host(gsmcal_ring* r, int slot) { return (r && slot >= 0 && slot < r->n) ? r->host[slot] : nullptr; }

int gsmcal_ring_submit(gsmcal_ring* r, int slot, size_t bytes) {
    if (!r || slot < 0 || slot >= r->n || bytes > r->bytes) return GSMCAL_E_ARG;
    gsmcal_ctx* c = r->c;
    ENTER(c);
    if (r->has_consumed[slot]) HIPCHK(c, hipStreamWaitEvent(r->copy, r->consumed[slot], 0));   // the device twin is free again
    HIPCHK(c, hipMemcpyAsync(r->dev[slot], r->host[slot], bytes ? bytes : r->bytes, hipMemcpyHostToDevice, r->copy));
    HIPCHK(c, hipEventRecord(r->copied[slot], r->copy));
    return 0;
}

void* gsmcal_ring_acquire(gsmcal_ring* r, int slot) {
    if (!r || slot < 0 || slot >= r->n) return nullptr;
    if (hipStreamWaitEvent(r->c->stream, r->copied[slot], 0) != hipSuccess) return nullptr;
    return r->dev[slot];
}

int gsmcal_ring_release(gsmcal_ring* r, int slot) {
    if (!r || slot < 0 || slot >= r->n) return GSMCAL_E_ARG;
    HIPCHK(r->c, hipEventRecord(r->consumed[slot], r->c->stream));
    r->has_consumed[slot] = 1;
    return 0;
}

int gsmcal_ring_host_ready(gsmcal_ring* r, int slot) {
    if (!r || slot < 0 || slot >= r->n) return GSMCAL_E_ARG;
    HIPCHK(r->c, hipEventSynchronize(r->copied[slot]));
    return 0;
}

