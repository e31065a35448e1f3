// Context, device memory, HIP-event timers and the RCCL communicator of libpermonhip.
#include <sched.h>

#include <algorithm>
#include <thread>

#include "pmh_internal.h"

static thread_local char g_err[1024] = "";

int pmh_set_error(int code, const char *fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char *pmh_last_error(void) { return g_err; }

// ---- run-time switches ---------------------------------------------------------------------------------------------------------
pmh_knobs_s &pmh_knobs()
{
  static pmh_knobs_s k = [] {
    pmh_knobs_s v;
    v.chain = getenv("PMH_NO_CHAIN") ? 0 : 1;
    v.svm_pairing = getenv("PMH_SVM_NO_PAIRING") ? 0 : 1;
    v.gt_fusion = getenv("PMH_NO_GT_FUSION") ? 0 : 1;
    v.multi_rhs = getenv("PMH_NO_MULTI_RHS") ? 0 : 1;
    v.kplus_mv = getenv("PMH_NO_KPLUS_MV") ? 0 : 1;
    v.smalxe_prefetch = getenv("PMH_SMALXE_NO_PREFETCH") ? 0 : 1;
    v.mg_d0_fusion = getenv("PMH_MG_NO_D0_FUSION") ? 0 : 1;
    v.vec_epi = getenv("PMH_NO_VEC_EPI") ? 0 : 1;
    v.mpgp_spec = getenv("PMH_MPGP_NO_SPEC") ? 0 : 1;
    int         nt = 0;
    const char *e  = getenv("PMH_HOST_THREADS");
    if (!e) e = getenv("OMP_NUM_THREADS");
    if (e) nt = atoi(e);
    if (nt <= 0) {
      cpu_set_t set;
      CPU_ZERO(&set);
      nt = (sched_getaffinity(0, sizeof(set), &set) == 0) ? CPU_COUNT(&set) : (int)std::thread::hardware_concurrency();
      nt = std::min(16, nt);
    }
    v.host_threads = std::max(1, std::min(64, nt));
    return v;
  }();
  return k;
}
static int *knob_by_name(const char *name)
{
  if (!name) return nullptr;
  if (!strcmp(name, "chain")) return &pmh_knobs().chain;
  if (!strcmp(name, "chain_applies")) return &pmh_knobs().chain_applies;
  if (!strcmp(name, "chain_launches")) return &pmh_knobs().chain_launches;
  if (!strcmp(name, "host_threads")) return &pmh_knobs().host_threads;
  if (!strcmp(name, "svm_pairing")) return &pmh_knobs().svm_pairing;
  if (!strcmp(name, "gt_fusion")) return &pmh_knobs().gt_fusion;
  if (!strcmp(name, "multi_rhs")) return &pmh_knobs().multi_rhs;
  if (!strcmp(name, "kplus_mv")) return &pmh_knobs().kplus_mv;
  if (!strcmp(name, "smalxe_prefetch")) return &pmh_knobs().smalxe_prefetch;
  if (!strcmp(name, "mg_d0_fusion")) return &pmh_knobs().mg_d0_fusion;
  if (!strcmp(name, "mpgp_spec")) return &pmh_knobs().mpgp_spec;
  if (!strcmp(name, "vec_epi")) return &pmh_knobs().vec_epi;
  return nullptr;
}
extern "C" int pmh_set_knob(const char *name, int value)
{
  int *k = knob_by_name(name);
  if (!k) return pmh_set_error(PMH_ERR_ARG, "pmh_set_knob: unknown switch '%s'", name ? name : "(null)");
  *k = value;
  return PMH_SUCCESS;
}
extern "C" int pmh_get_knob(const char *name, int *value)
{
  int *k = knob_by_name(name);
  if (!k || !value) return pmh_set_error(PMH_ERR_ARG, "pmh_get_knob: unknown switch '%s'", name ? name : "(null)");
  *value = *k;
  return PMH_SUCCESS;
}

extern "C" int pmh_init(int device, pmh_ctx *out)
{
  PMH_ARG(out);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return pmh_set_error(PMH_ERR_NODEVICE, "pmh_init: no HIP device visible (libpermonhip has no CPU fallback)");
  if (device < 0 || device >= ndev) return pmh_set_error(PMH_ERR_ARG, "pmh_init: device %d out of range [0,%d)", device, ndev);
  PMH_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  PMH_HIP(hipGetDeviceProperties(&prop, device));
  if (!strstr(prop.gcnArchName, "gfx950")) return pmh_set_error(PMH_ERR_NODEVICE, "pmh_init: device %d is %s; libpermonhip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  pmh_ctx c  = new pmh_ctx_s();
  c->device  = device;
  c->num_cus = prop.multiProcessorCount;
  c->comm    = nullptr;
  c->rank    = 0;
  c->size    = 1;
  c->dist_scalars = 0;
  c->force_comm = getenv("PMH_COMM_FORCE") ? 1 : 0; // testing: keep the RCCL calls on a 1-rank communicator
  PMH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  PMH_HIP(hipEventCreate(&c->ev0));
  PMH_HIP(hipEventCreate(&c->ev1));
  c->partials_cap = PMH_MAX_VEC_BLOCKS;
  PMH_HIP(hipMalloc((void **)&c->d_partials, sizeof(double) * PMH_MAX_RED * c->partials_cap));
  PMH_HIP(hipMalloc((void **)&c->d_scal, sizeof(double) * PMH_NSCAL));
  PMH_HIP(hipMemset(c->d_scal, 0, sizeof(double) * PMH_NSCAL));
  PMH_HIP(hipHostMalloc((void **)&c->h_scal, sizeof(double) * PMH_NSCAL, hipHostMallocMapped));
  memset(c->h_scal, 0, sizeof(double) * PMH_NSCAL);
  PMH_HIP(hipHostMalloc((void **)&c->h_partials, sizeof(double) * PMH_MAX_RED * c->partials_cap, hipHostMallocMapped));
  memset(c->h_partials, 0, sizeof(double) * PMH_MAX_RED * c->partials_cap);
  PMH_HIP(hipMalloc((void **)&c->d_commbuf, sizeof(double) * PMH_NSCAL));
  *out = c;
  return PMH_SUCCESS;
}

extern "C" int pmh_finalize(pmh_ctx c)
{
  if (!c) return PMH_SUCCESS;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  if (c->comm) ncclCommDestroy(c->comm);
  for (int i = 0; i < c->comm_ev_cap; i++) hipEventDestroy(c->comm_ev[i]);
  delete[] c->comm_ev;
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  hipFree(c->d_partials);
  hipFree(c->d_scal);
  hipFree(c->d_commbuf);
  hipHostFree(c->h_scal);
  hipHostFree(c->h_partials);
  hipEventDestroy(c->ev0);
  hipEventDestroy(c->ev1);
  hipStreamDestroy(c->stream);
  delete c;
  return PMH_SUCCESS;
}

extern "C" int pmh_device_name(pmh_ctx c, char *buf, size_t len)
{
  PMH_ARG(c && buf && len > 0);
  hipDeviceProp_t prop;
  PMH_HIP(hipGetDeviceProperties(&prop, c->device));
  snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return PMH_SUCCESS;
}

// free / total bytes of the device's HBM (hipMemGetInfo): sizing against the 288 GB, leak checks in the tests
extern "C" int pmh_mem_info(pmh_ctx c, size_t *free_bytes, size_t *total_bytes)
{
  PMH_ARG(c && free_bytes && total_bytes);
  PMH_HIP(hipSetDevice(c->device));
  PMH_HIP(hipStreamSynchronize(c->stream));
  PMH_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return PMH_SUCCESS;
}

// the host waits for the launch stream once per MPGP step (step decisions on a shell operator).  (Polling the stream instead of blocking in hipStreamSynchronize was a
// knob until round 6: measured, noise.)
static inline hipError_t ctx_wait(pmh_ctx c) { return hipStreamSynchronize(c->stream); }

extern "C" int pmh_sync(pmh_ctx c)
{
  PMH_ARG(c);
  PMH_HIP(ctx_wait(c));
  return PMH_SUCCESS;
}

extern "C" void *pmh_stream(pmh_ctx c) { return c ? (void *)c->stream : nullptr; }

extern "C" int pmh_malloc(pmh_ctx c, size_t bytes, void **dptr)
{
  PMH_ARG(c && dptr);
  PMH_HIP(hipSetDevice(c->device));
  PMH_HIP(hipMalloc(dptr, bytes ? bytes : 8));
  return PMH_SUCCESS;
}

extern "C" int pmh_free(pmh_ctx c, void *dptr)
{
  PMH_ARG(c);
  if (dptr) {
    PMH_HIP(hipStreamSynchronize(c->stream));
    PMH_HIP(hipFree(dptr));
  }
  return PMH_SUCCESS;
}

extern "C" int pmh_memcpy_h2d(pmh_ctx c, void *dst, const void *src, size_t bytes)
{
  PMH_ARG(c);
  if (!bytes) return PMH_SUCCESS;
  PMH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  PMH_HIP(hipStreamSynchronize(c->stream));
  return PMH_SUCCESS;
}

extern "C" int pmh_memcpy_d2h(pmh_ctx c, void *dst, const void *src, size_t bytes)
{
  PMH_ARG(c);
  if (!bytes) return PMH_SUCCESS;
  PMH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  PMH_HIP(hipStreamSynchronize(c->stream));
  return PMH_SUCCESS;
}

extern "C" int pmh_memcpy_d2d(pmh_ctx c, void *dst, const void *src, size_t bytes)
{
  PMH_ARG(c);
  if (!bytes || dst == src) return PMH_SUCCESS;
  PMH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
  return PMH_SUCCESS;
}

extern "C" int pmh_memset(pmh_ctx c, void *dst, int value, size_t bytes)
{
  PMH_ARG(c);
  if (!bytes) return PMH_SUCCESS;
  PMH_HIP(hipMemsetAsync(dst, value, bytes, c->stream));
  return PMH_SUCCESS;
}

extern "C" int pmh_timer_start(pmh_ctx c)
{
  PMH_ARG(c);
  PMH_HIP(hipEventRecord(c->ev0, c->stream));
  return PMH_SUCCESS;
}

extern "C" int pmh_timer_stop(pmh_ctx c, double *ms)
{
  PMH_ARG(c && ms);
  float f = 0.f;
  PMH_HIP(hipEventRecord(c->ev1, c->stream));
  PMH_HIP(hipEventSynchronize(c->ev1));
  PMH_HIP(hipEventElapsedTime(&f, c->ev0, c->ev1));
  *ms = (double)f;
  return PMH_SUCCESS;
}

// ---- RCCL over xGMI ---------------------------------------------------------------------------------------
extern "C" int pmh_comm_unique_id(void *id128)
{
  PMH_ARG(id128);
  static_assert(sizeof(ncclUniqueId) <= PMH_UNIQUE_ID_BYTES, "ncclUniqueId larger than PMH_UNIQUE_ID_BYTES");
  ncclUniqueId id;
  PMH_NCCL(ncclGetUniqueId(&id));
  memset(id128, 0, PMH_UNIQUE_ID_BYTES);
  memcpy(id128, &id, sizeof(id));
  return PMH_SUCCESS;
}

extern "C" int pmh_comm_init(pmh_ctx c, int rank, int size, const void *id128)
{
  PMH_ARG(c && id128 && size >= 1 && rank >= 0 && rank < size);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  PMH_HIP(hipSetDevice(c->device));
  PMH_NCCL(ncclCommInitRank(&c->comm, size, id, rank));
  c->rank = rank;
  c->size = size;
  return PMH_SUCCESS;
}

extern "C" int pmh_comm_rank(pmh_ctx c, int *rank, int *size)
{
  PMH_ARG(c);
  if (rank) *rank = c->rank;
  if (size) *size = c->size;
  return PMH_SUCCESS;
}

// Host-staged transport of the collectives: device -> pinned host -> fn (an in-place all-reduce over the ranks, e.g. MPI_Allreduce(MPI_IN_PLACE, ...) on the
// communicator of the PETSc objects the glue was handed) -> device, in stream order.  For deployments whose ranks cannot form an RCCL communicator (several
// ranks per GPU, a fabric RCCL does not drive) and for the 2-process tests on a one-GPU box; RCCL over xGMI (pmh_comm_init) stays the transport of the
// benchmarked path.
extern "C" int pmh_comm_set_host_transport(pmh_ctx c, int rank, int size, pmh_comm_host_fn fn, void *user)
{
  PMH_ARG(c && size >= 1 && rank >= 0 && rank < size);
  if (fn && c->comm) return pmh_set_error(PMH_ERR_STATE, "pmh_comm_set_host_transport: the context already has an RCCL communicator");
  const bool had_hook = c->hook != nullptr;
  c->hook = fn, c->hook_user = user;
  if (fn) c->rank = rank, c->size = size;
  // removing the hook of a context that had nothing else: one rank again (an RCCL communicator keeps its rank / size)
  else if (had_hook && !c->comm) c->rank = 0, c->size = 1;
  return PMH_SUCCESS;
}

static int host_reduce(pmh_ctx c, int op, double *dbuf, size_t count)
{
  if (count > c->stage_cap) {
    if (c->h_stage) PMH_HIP(hipHostFree(c->h_stage));
    c->h_stage = nullptr, c->stage_cap = 0;
    PMH_HIP(hipHostMalloc((void **)&c->h_stage, sizeof(double) * std::max<size_t>(count, 1024), hipHostMallocDefault));
    c->stage_cap = std::max<size_t>(count, 1024);
  }
  if (count) PMH_HIP(hipMemcpyAsync(c->h_stage, dbuf, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
  PMH_HIP(hipStreamSynchronize(c->stream));
  if (const int rc = c->hook(c->hook_user, op, c->h_stage, count)) return pmh_set_error(PMH_ERR_COMM, "host transport: the all-reduce callback returned %d", rc);
  // (the next staging copy is ordered behind it on the stream)
  if (count) PMH_HIP(hipMemcpyAsync(dbuf, c->h_stage, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
  return PMH_SUCCESS;
}

extern "C" int pmh_comm_allreduce_sum(pmh_ctx c, double *dbuf, size_t count)
{
  PMH_ARG(c);
  if (!pmh_comm_on(c) || !count) return PMH_SUCCESS;
  // optional timing of the data-path exchanges (vectors, not scalars): HIP event pairs on the launch stream (pmh_comm_timing_enable)
  const bool timed = c->comm_ev_cap > 0 && count >= 1024 && 2 * (c->comm_ev_used + 1) <= c->comm_ev_cap;
  if (timed) PMH_HIP(hipEventRecord(c->comm_ev[2 * c->comm_ev_used], c->stream));
  if (c->hook) PMH_CHK(host_reduce(c, PMH_COMM_SUM, dbuf, count));
  else PMH_NCCL(ncclAllReduce(dbuf, dbuf, count, ncclDouble, ncclSum, c->comm, c->stream));
  if (timed) {
    PMH_HIP(hipEventRecord(c->comm_ev[2 * c->comm_ev_used + 1], c->stream));
    c->comm_ev_used++;
    c->comm_ev_bytes += 8.0 * (double)count;
  }
  return PMH_SUCCESS;
}

// event pairs around the next max_events vector all-reduces (0: off, frees the events); pmh_comm_timing_get: how many were timed, their total milliseconds and
// bytes
extern "C" int pmh_comm_timing_enable(pmh_ctx c, int max_events)
{
  PMH_ARG(c && max_events >= 0);
  PMH_HIP(hipStreamSynchronize(c->stream));
  for (int i = 0; i < c->comm_ev_cap; i++) hipEventDestroy(c->comm_ev[i]);
  delete[] c->comm_ev;
  c->comm_ev = nullptr, c->comm_ev_cap = 0, c->comm_ev_used = 0, c->comm_ev_bytes = 0.0;
  if (max_events > 0) {
    c->comm_ev = new hipEvent_t[2 * (size_t)max_events];
    for (int i = 0; i < 2 * max_events; i++) PMH_HIP(hipEventCreate(&c->comm_ev[i]));
    c->comm_ev_cap = 2 * max_events;
  }
  return PMH_SUCCESS;
}
extern "C" int pmh_comm_timing_get(pmh_ctx c, int *count, double *total_ms, double *bytes)
{
  PMH_ARG(c && count && total_ms);
  PMH_HIP(hipStreamSynchronize(c->stream));
  *count = c->comm_ev_used, *total_ms = 0.0;
  for (int i = 0; i < c->comm_ev_used; i++) {
    float ms = 0.f;
    PMH_HIP(hipEventElapsedTime(&ms, c->comm_ev[2 * i], c->comm_ev[2 * i + 1]));
    *total_ms += ms;
  }
  if (bytes) *bytes = c->comm_ev_bytes;
  c->comm_ev_used = 0, c->comm_ev_bytes = 0.0;
  return PMH_SUCCESS;
}

extern "C" int pmh_comm_allreduce_min(pmh_ctx c, double *dbuf, size_t count)
{
  PMH_ARG(c);
  if (!pmh_comm_on(c) || !count) return PMH_SUCCESS;
  if (c->hook) return host_reduce(c, PMH_COMM_MIN, dbuf, count);
  PMH_NCCL(ncclAllReduce(dbuf, dbuf, count, ncclDouble, ncclMin, c->comm, c->stream));
  return PMH_SUCCESS;
}

// K device scalars with their own operations (the MPGP reductions of a phase: sums and QPCFeas's MIN) in one grouped exchange
int pmh_comm_allreduce_scalars(pmh_ctx c, double *dscal, int K, const int *ops)
{
  if (!pmh_comm_on(c) || K <= 0) return PMH_SUCCESS;
  if (c->hook) {
    // runs of equal operations go out together (same values as one exchange per scalar: an all-reduce acts element-wise)
    for (int k0 = 0; k0 < K;) {
      int k1 = k0 + 1;
      while (k1 < K && (ops[k1] == PMH_RED_MIN) == (ops[k0] == PMH_RED_MIN)) k1++;
      PMH_CHK(host_reduce(c, ops[k0] == PMH_RED_MIN ? PMH_COMM_MIN : PMH_COMM_SUM, dscal + k0, (size_t)(k1 - k0)));
      k0 = k1;
    }
    return PMH_SUCCESS;
  }
  PMH_NCCL(ncclGroupStart());
  for (int k = 0; k < K; k++) PMH_NCCL(ncclAllReduce(dscal + k, dscal + k, 1, ncclDouble, ops[k] == PMH_RED_MIN ? ncclMin : ncclSum, c->comm, c->stream));
  PMH_NCCL(ncclGroupEnd());
  return PMH_SUCCESS;
}

extern "C" int pmh_comm_barrier(pmh_ctx c)
{
  PMH_ARG(c);
  if (pmh_comm_on(c)) {
    if (c->hook) PMH_CHK(host_reduce(c, PMH_COMM_BARRIER, c->d_commbuf, 0));
    else PMH_NCCL(ncclAllReduce(c->d_commbuf, c->d_commbuf, 1, ncclDouble, ncclSum, c->comm, c->stream));
  }
  PMH_HIP(hipStreamSynchronize(c->stream));
  return PMH_SUCCESS;
}

int pmh_host_scalar(pmh_ctx c, int slot, double *v)
{
  PMH_HIP(ctx_wait(c));
  *v = c->h_scal[slot];
  return PMH_SUCCESS;
}
