// hip_backend.hip -- the HIP side of libfnft_amd.so: the extern "C" shim (sections 2 and 3 of
// include/fnft_amd.h plus the internal entries the C drivers call) over NftPlan<HipBackend>.  The kernels
// are instantiated in hip_kernels_*.hip (see hip_be.h).  gfx950 only.
#include <map>
#include <mutex>
#include <tuple>

#include "hip_be.h"
#include "nft_nsev_inverse.h"
#include "../../include/fnft_amd.h"

thread_local std::string g_last_error;
// fnft__misc.c:379-380
extern "C" void fnft_amd__warn(const char *msg, const char *func, int line);   // fnft_nsev_host.c
static const char *const kNotBandlimitedMsg =
    "Signal does not appear to be bandlimited. Interpolation step may be inaccurate. Try to reduce the step size, "
    "or switch to a discretization that does not require interpolation";
// Host-pointer entry points share cached plans and workspaces PER DEVICE: one call at a time on a device (SURVEY 8b
// allows an internal mutex), calls on different devices run concurrently; device-resident plans are independent
// objects with their own lock.
static std::mutex &host_call_mutex_of(int d)
{
    static std::mutex m[64];
    return m[(unsigned)d % 64u];
}
static std::mutex &host_call_mutex()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return host_call_mutex_of(d);
}

using Plan = NftPlan<HipBackend>;

// ---- cache of released device blocks (HipBackend::alloc / free) ------------------------------------------------------
// Sizes are rounded up to a bucket (powers of two below 1 MiB, four steps per octave above: at most 25 % slack) so
// that the work arrays of a repeated call find the blocks of the previous one.  At most kPoolMaxCached bytes per
// process stay cached; fnft_amd_release_cached() returns everything to the driver.
namespace {
struct DevPool {
    std::mutex m;
    std::multimap<std::pair<int, size_t>, void *> idle;    // (device, bucket) -> block
    std::map<void *, std::pair<int, size_t>> live;         // block -> (device, bucket)
    size_t cached = 0;
    static constexpr size_t kPoolMaxCached = (size_t)6 << 30;
    static size_t bucket(size_t b)
    {
        if (b < 256) b = 256;
        size_t p = 256;
        while (p < b) p *= 2;
        if (p <= ((size_t)1 << 20)) return p;
        const size_t step = p / 8;          // p/2 < b <= p: buckets p/2 + k*p/8
        return (b + step - 1) / step * step;
    }
    void trim_locked(int device)
    {
        for (auto it = idle.begin(); it != idle.end();) {
            if (device < 0 || it->first.first == device) {
                cached -= it->first.second;
                (void)hipFree(it->second);
                it = idle.erase(it);
            } else ++it;
        }
    }
};
DevPool &pool() { static DevPool *p = new DevPool(); return *p; }   // never destroyed: blocks outlive static teardown
}  // namespace

void *fa_pool_alloc(size_t bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { g_last_error = "hipGetDevice failed"; return nullptr; }
    DevPool &P = pool();
    const size_t bk = DevPool::bucket(bytes);
    std::lock_guard<std::mutex> lk(P.m);
    auto it = P.idle.find(std::make_pair(dev, bk));
    void *p = nullptr;
    if (it != P.idle.end()) {
        p = it->second;
        P.idle.erase(it);
        P.cached -= bk;
    } else {
        if (hipMalloc(&p, bk) != hipSuccess) {
            (void)hipGetLastError();
            P.trim_locked(dev);   // out of memory: give the cached blocks back and try once more
            if (!hip_ok(hipMalloc(&p, bk), "hipMalloc")) return nullptr;
        }
    }
    P.live[p] = std::make_pair(dev, bk);
    return p;
}

void fa_pool_free(void *p)
{
    DevPool &P = pool();
    std::lock_guard<std::mutex> lk(P.m);
    auto it = P.live.find(p);
    if (it == P.live.end()) { (void)hipFree(p); return; }
    const std::pair<int, size_t> key = it->second;
    P.live.erase(it);
    if (P.cached + key.second > DevPool::kPoolMaxCached) { (void)hipFree(p); return; }
    P.idle.insert(std::make_pair(key, p));
    P.cached += key.second;
}

struct fnft_amd_plan {
    HipBackend be;
    Plan *pl = nullptr;
    int device = 0;
    int nse_disc = 0;
    int kdv_disc = -1;   // >= 0: plan made by fnft_amd_kdvv_plan_create
    int real_mode = -1;  // KdV plans: -1 ask the device whether the potential is real (default), 0 complex path, 1 real path
    std::mutex mtx;
};

// Device rule of the library (documented in include/fnft_amd.h): the host-pointer entry points
// compute on the calling thread's CURRENT HIP device (hipGetDevice; what torch.cuda.set_device or
// hipSetDevice selected) and never change it; device-resident plans live on the device given to
// fnft_amd_plan_create, and every plan call restores the caller's current device before returning.
static bool device_valid(int device)
{
    int n = 0;
    if (!hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount") || n <= 0) {
        if (g_last_error.empty()) g_last_error = "no HIP device";
        return false;
    }
    if (device < 0 || device >= n) { g_last_error = "device index out of range"; return false; }
    return true;
}
// the calling thread's current device, or -1 (no usable device)
static int current_device()
{
    int n = 0, dev = 0;
    if (!hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount") || n <= 0) {
        if (g_last_error.empty()) g_last_error = "no HIP device";
        return -1;
    }
    if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return -1;
    return dev;
}
// selects `device` for the scope and puts the caller's device back afterwards
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device)
    {
        if (!device_valid(device)) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (prev == device) || hip_ok(hipSetDevice(device), "hipSetDevice");
    }
    ~DeviceGuard()
    {
        if (ok && prev >= 0) {
            int now = -1;
            if (hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
        }
    }
};

// state the host-pointer entry points keep between calls, per device (fnft_amd_release_cached gives it back)
struct Peeler {
    HipBackend be;
    NftLayerPeelingDev<HipBackend> lp;
    Peeler() : lp(be, 1.0, 1, 1) {}
};
static std::map<int, Peeler *> g_peelers;
static std::mutex g_peelers_mtx;   // the map itself; a device's peeler is used under that device's host-call lock
static std::mutex g_cache_mtx;     // the two plan caches below
static std::map<std::tuple<int, size_t, size_t, int, size_t>, fnft_amd_plan *> g_nsev_cache;   // (dev, D, M, disc, nskip)
static std::map<std::tuple<int, size_t, size_t, int>, fnft_amd_plan *> g_kdvv_cache;           // (dev, D, M, disc)

extern "C" {

int fnft_amd_device_count(void)
{
    int n = 0;
    if (!hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount")) return -1;
    return n;
}

const char *fnft_amd_last_error(void) { return g_last_error.c_str(); }

FNFT_INT fnft_amd_plan_create(fnft_amd_plan_t **plan, FNFT_UINT D, FNFT_UINT M, FNFT_UINT batch,
                              fnft_nse_discretization_t discretization, int device)
{
    return fnft_amd_plan_create_sub(plan, D, M, batch, discretization, device, 1);
}

FNFT_INT fnft_amd_plan_create_sub(fnft_amd_plan_t **plan, FNFT_UINT D, FNFT_UINT M, FNFT_UINT batch,
                                  fnft_nse_discretization_t discretization, int device, FNFT_UINT nskip)
{
    if (!plan || D < 2 || batch < 1 || nskip < 1) return FNFT_EC_INVALID_ARGUMENT;
    const int akns = nft_nse_to_akns((int)discretization);
    if (akns < 0) return FNFT_EC_NOT_YET_IMPLEMENTED;
    const int ups = nft_nse_upsampling((int)discretization);
    if (ups == 1 && nskip != 1) return FNFT_EC_NOT_YET_IMPLEMENTED;
    if (ups == 2 && D <= 2) return FNFT_EC_INVALID_ARGUMENT;   // fnft__misc.c:331-332
    DeviceGuard dg(device);
    if (!dg.ok) return FNFT_EC_OTHER;
    fnft_amd_plan *P = new (std::nothrow) fnft_amd_plan();
    if (!P) return FNFT_EC_NOMEM;
    P->device = device;
    P->nse_disc = (int)discretization;
    const size_t Dtree = (ups == 1) ? (size_t)D : 2 * Plan::sub_count((size_t)D, (size_t)nskip);
    P->pl = new (std::nothrow) Plan(P->be, Dtree, M, batch, akns, nft_akns_degree(akns));
    if (!P->pl) { delete P; return FNFT_EC_NOMEM; }
    P->pl->set_front((size_t)D, (size_t)nskip, ups);
#ifdef FNFT_AMD_ABLATION   // diagnostic builds only; the product library reads no environment variable
    if (const char *dbg = getenv("FNFT_AMD_DBG")) P->pl->dbg_flags = atoi(dbg);
#endif
    const int rc = P->pl->init();
    if (rc != NFT_SUCCESS || P->be.failed) {
        P->pl->destroy();
        delete P->pl;
        delete P;
        return rc != NFT_SUCCESS ? rc : FNFT_EC_NOMEM;
    }
    (void)P->be.sync();
    *plan = P;
    return FNFT_SUCCESS;
}

void fnft_amd_plan_destroy(fnft_amd_plan_t *plan)
{
    if (!plan) return;
    DeviceGuard dg(plan->device);
    (void)hipDeviceSynchronize();
    plan->pl->destroy();
    plan->be.destroy_events();
    delete plan->pl;
    delete plan;
}

FNFT_UINT fnft_amd_plan_workspace_bytes(const fnft_amd_plan_t *plan) { return plan ? plan->pl->bytes : 0; }

int fnft_amd_plan_device(const fnft_amd_plan_t *plan) { return plan ? plan->device : -1; }

int fnft_amd_plan_last_warnings(const fnft_amd_plan_t *plan) { return plan ? plan->pl->last_warn : 0; }

// host wall-clock stages of the calling thread's last discrete-spectrum call (measurement only)
double fnft_amd_discspec_stage_ms(FNFT_UINT i, char *name, FNFT_UINT name_cap)
{
    const std::vector<NftDsStage> &v = nft_ds_stages();
    if (i >= v.size()) return -1.0;
    if (name && name_cap) {
        size_t n = strlen(v[i].what);
        if (n >= name_cap) n = name_cap - 1;
        memcpy(name, v[i].what, n);
        name[n] = 0;
    }
    return v[i].ms;
}

#ifdef FNFT_AMD_TUNING
// diagnostic builds only: tuning parameters of a plan (which: 0 = row-kernel stagger)
extern "C" int fnft_amd_debug_tune(fnft_amd_plan_t *plan, int which, int value)
{
    if (!plan) return -1;
    if (which == 0) plan->pl->tune_stagger = value;
    return 0;
}
#endif

#ifdef FNFT_AMD_STAMPS
// diagnostic builds only: stamp the row kernel of split level `level` (16 u64 per wave, 4096 waves max);
// read back after the stream has been synchronised
extern "C" int fnft_amd_debug_stamps(fnft_amd_plan_t *plan, int level, unsigned long long *host, size_t n)
{
    if (!plan) return -1;
    DeviceGuard dg(plan->device);
    Plan &pl = *plan->pl;
    if (!pl.dbg_stamps) {
        pl.dbg_stamps = (unsigned long long *)plan->be.alloc((size_t)4096 * 16 * 8);
        if (!pl.dbg_stamps) return -1;
        (void)hipMemset(pl.dbg_stamps, 0, (size_t)4096 * 16 * 8);
    }
    pl.stamp_level = level;
    if (host && n) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(host, pl.dbg_stamps, n * 8, hipMemcpyDeviceToHost);
    }
    return 0;
}
#endif

int fnft_amd_current_device(void) { return current_device(); }

void fnft_amd_plan_set_timing(fnft_amd_plan_t *plan, int enabled)
{
    if (plan) plan->be.timing = enabled != 0;
}

double fnft_amd_plan_last_ms(const fnft_amd_plan_t *plan, int which)
{
    if (!plan) return -1.0;
    switch (which) {
    case 0: return plan->be.elapsed_ms(0, 1);
    case 1: return plan->be.elapsed_ms(1, 2);
    case 2: return plan->be.elapsed_ms(0, 2);
    default: return -1.0;
    }
}

void fnft_amd_plan_set_launch_timing(fnft_amd_plan_t *plan, int enabled)
{
    if (!plan) return;
    plan->be.launch_timing = enabled != 0;
    plan->be.launches_used = 0;
}

FNFT_UINT fnft_amd_plan_launch_count(const fnft_amd_plan_t *plan)
{
    return plan ? plan->be.launches_used : 0;
}

double fnft_amd_plan_launch_ms(const fnft_amd_plan_t *plan, FNFT_UINT i, char *name, FNFT_UINT name_cap)
{
    if (!plan || i >= plan->be.launches_used) return -1.0;
    const HipBackend::LaunchRec &r = plan->be.launches[i];
    if (name && name_cap) {
        // __PRETTY_FUNCTION__ of run<K>: "... [K = KMid<2>]"
        const char *k = strstr(r.name, "K = ");
        k = k ? k + 4 : r.name;
        size_t n = strlen(k);
        if (n && k[n - 1] == ']') n--;
        if (n >= name_cap) n = name_cap - 1;
        memcpy(name, k, n);
        name[n] = 0;
    }
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return -1.0;
    return (double)ms;
}

FNFT_INT fnft_amd_nsev_contspec_device(fnft_amd_plan_t *plan, const void *d_q, void *d_contspec,
                                       const FNFT_REAL *T, const FNFT_REAL *XI, FNFT_INT kappa,
                                       fnft_nsev_cstype_t contspec_type,
                                       FNFT_INT normalization_flag, void *stream)
{
    if (!plan || !d_q || !T || !(T[0] < T[1])) return FNFT_EC_INVALID_ARGUMENT;
    if (d_contspec && (!XI || !(XI[0] < XI[1]))) return FNFT_EC_INVALID_ARGUMENT;
    if (kappa != 1 && kappa != -1) return FNFT_EC_INVALID_ARGUMENT;
    const int cst = (int)contspec_type;
    if (cst < 0 || cst > 2) return FNFT_EC_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(plan->mtx);
    DeviceGuard dg(plan->device);
    if (!dg.ok) return FNFT_EC_OTHER;
    Plan &pl = *plan->pl;
    plan->be.stream = (hipStream_t)stream;
    plan->be.failed = false;
    plan->be.mark(0);
    double Tsub[2];
    int rc = pl.run_front(d_q, T, kappa, Tsub);
    if (rc == NFT_SUCCESS) rc = pl.run_tree();
    plan->be.mark(1);
    if (rc == NFT_SUCCESS && d_contspec && pl.M > 0) {
        Plan::Contspec cs;
        cs.T[0] = Tsub[0]; cs.T[1] = Tsub[1]; cs.XI[0] = XI[0]; cs.XI[1] = XI[1];
        cs.nse_disc = plan->nse_disc;
        cs.cstype = cst;
        cs.normalization_flag = normalization_flag;
        rc = pl.run_contspec(d_contspec, cs);
    }
    plan->be.mark(2);
    if (plan->be.failed) return FNFT_EC_OTHER;
    return rc;
}

// nsev_compute_contspec (src/fnft_nsev.c:744-891) on a caller-supplied transfer matrix in device memory
FNFT_INT fnft_amd_nsev_contspec_from_tm_device(fnft_amd_plan_t *plan, const void *d_tm, FNFT_INT W, void *d_contspec,
                                               const FNFT_REAL *T, const FNFT_REAL *XI,
                                               fnft_nsev_cstype_t contspec_type, void *stream)
{
    if (!plan || plan->kdv_disc >= 0 || !d_tm || !d_contspec || !T || !(T[0] < T[1]) || !XI || !(XI[0] < XI[1]))
        return FNFT_EC_INVALID_ARGUMENT;
    const int cst = (int)contspec_type;
    if (cst < 0 || cst > 2 || plan->pl->M == 0) return FNFT_EC_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(plan->mtx);
    DeviceGuard dg(plan->device);
    if (!dg.ok) return FNFT_EC_OTHER;
    Plan &pl = *plan->pl;
    plan->be.stream = (hipStream_t)stream;
    plan->be.failed = false;
    Plan::Contspec cs;
    cs.T[0] = T[0]; cs.T[1] = T[1]; cs.XI[0] = XI[0]; cs.XI[1] = XI[1];
    cs.nse_disc = plan->nse_disc;
    cs.cstype = cst;
    cs.normalization_flag = 1;
    const int rc = pl.run_contspec_tm(d_contspec, cs, d_tm, (int)W);
    if (plan->be.failed) return FNFT_EC_OTHER;
    return rc;
}

// ---- KdV ----------------------------------------------------------------------------------------
FNFT_INT fnft_amd_kdvv_plan_create(fnft_amd_plan_t **plan, FNFT_UINT D, FNFT_UINT M, FNFT_UINT batch,
                                   fnft_kdv_discretization_t discretization, int device)
{
    if (!plan || D < 2 || batch < 1 || M < 1) return FNFT_EC_INVALID_ARGUMENT;
    const int kd = (int)discretization;
    if (kd < 0) return FNFT_EC_INVALID_ARGUMENT;
    if (kd > (int)fnft_kdv_discretization_2SPLIT8B) return FNFT_EC_NOT_YET_IMPLEMENTED;
    const int akns = kd + 1;   // same scheme names, fnft__kdv_discretization.c:86-150
    DeviceGuard dg(device);
    if (!dg.ok) return FNFT_EC_OTHER;
    fnft_amd_plan *P = new (std::nothrow) fnft_amd_plan();
    if (!P) return FNFT_EC_NOMEM;
    P->device = device;
    P->nse_disc = -1;
    P->kdv_disc = kd;
    P->pl = new (std::nothrow) Plan(P->be, D, M, batch, akns, nft_akns_degree(akns));
    if (!P->pl) { delete P; return FNFT_EC_NOMEM; }
    P->pl->kdv = true;
    const int rc = P->pl->init();
    if (rc != NFT_SUCCESS || P->be.failed) {
        P->pl->destroy();
        delete P->pl;
        delete P;
        return rc != NFT_SUCCESS ? rc : FNFT_EC_NOMEM;
    }
    (void)P->be.sync();
    *plan = P;
    return FNFT_SUCCESS;
}

FNFT_INT fnft_amd_kdvv_contspec_device(fnft_amd_plan_t *plan, const void *d_u, void *d_contspec,
                                       const FNFT_REAL *T, const FNFT_REAL *XI, void *stream)
{
    if (!plan || plan->kdv_disc < 0 || !d_u || !d_contspec || !T || !(T[0] < T[1]) || !XI || !(XI[0] < XI[1]))
        return FNFT_EC_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(plan->mtx);
    DeviceGuard dg(plan->device);
    if (!dg.ok) return FNFT_EC_OTHER;
    Plan &pl = *plan->pl;
    plan->be.stream = (hipStream_t)stream;
    plan->be.failed = false;
    plan->be.mark(0);
    // a real potential takes the real-coefficient tree (nft_real.h: half the transforms and bytes); in the default mode
    // the device is asked first -- one small kernel and a 4-byte read-back, i.e. this call then waits for the stream once
    bool real = plan->real_mode == 1;
    if (plan->real_mode < 0) {
        RealCheckParams R;
        R.q = (const cplx *)d_u;
        R.n = (long long)(pl.batch * pl.D);
        R.flag = pl.status + 1;
        plan->be.run<KRealCheck>((int)std::min<long long>(1024, (R.n + 255) / 256), 1, R);
        int flag = 1;
        plan->be.d2h(&flag, pl.status + 1, sizeof(int));
        if (plan->be.sync() != NFT_SUCCESS) return FNFT_EC_OTHER;
        if (flag) plan->be.memset0(pl.status + 1, sizeof(int));
        real = (flag == 0);
    }
    pl.want_real = real;
    double Tsub[2];
    int rc = pl.run_front(d_u, T, 1, Tsub);
    if (rc == NFT_SUCCESS) rc = pl.run_tree();
    plan->be.mark(1);
    if (rc == NFT_SUCCESS)
        rc = pl.run_contspec_kdv(d_contspec, T, XI, plan->kdv_disc == (int)fnft_kdv_discretization_2SPLIT2A);
    plan->be.mark(2);
    if (plan->be.failed) return FNFT_EC_OTHER;
    return rc;
}

FNFT_INT fnft_amd_kdvv_plan_set_real_mode(fnft_amd_plan_t *plan, int mode)
{
    if (!plan || plan->kdv_disc < 0 || mode < -1 || mode > 1) return FNFT_EC_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(plan->mtx);
    plan->real_mode = mode;
    return FNFT_SUCCESS;
}

FNFT_INT fnft_amd_plan_finish(fnft_amd_plan_t *plan, void *stream)
{
    if (!plan) return FNFT_EC_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(plan->mtx);
    DeviceGuard dg(plan->device);
    plan->be.stream = (hipStream_t)stream;
    return plan->pl->read_status();
}

FNFT_INT fnft_amd_plan_get_transfer_matrix(fnft_amd_plan_t *plan, FNFT_UINT b,
                                           FNFT_COMPLEX *result_host, FNFT_UINT *deg, FNFT_INT *W)
{
    if (!plan || !result_host || !plan->pl->tree_valid || b >= plan->pl->batch)
        return FNFT_EC_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(plan->mtx);
    DeviceGuard dg(plan->device);
    Plan &pl = *plan->pl;
    pl.export_tm();
    const size_t per = 4 * (pl.res_deg + 1);
    plan->be.d2h(result_host, pl.tm_out + b * per, per * sizeof(cplx));
    int w = 0;
    plan->be.d2h(&w, pl.wexp[pl.cur] + b, sizeof(int));
    const int rc = plan->be.sync();
    if (deg) *deg = pl.res_deg;
    if (W) *W = w;
    return rc;
}

// Device variant: the transfer matrix of signal b goes to a DEVICE buffer of 4*(deg+1) complex128 (same layout),
// enqueued on `stream`; *deg and *W are returned to the host (this call waits for the stream: W is read back).
FNFT_INT fnft_amd_plan_get_transfer_matrix_device(fnft_amd_plan_t *plan, FNFT_UINT b, void *d_result,
                                                  FNFT_UINT *deg, FNFT_INT *W, void *stream)
{
    if (!plan || !d_result || !plan->pl->tree_valid || b >= plan->pl->batch) return FNFT_EC_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lk(plan->mtx);
    DeviceGuard dg(plan->device);
    if (!dg.ok) return FNFT_EC_OTHER;
    Plan &pl = *plan->pl;
    plan->be.stream = (hipStream_t)stream;
    pl.export_tm();
    const size_t per = 4 * (pl.res_deg + 1);
    if (!hip_ok(hipMemcpyAsync(d_result, pl.tm_out + b * per, per * sizeof(cplx), hipMemcpyDeviceToDevice,
                               plan->be.stream), "hipMemcpyAsync(D2D)"))
        return FNFT_EC_OTHER;
    int w = 0;
    plan->be.d2h(&w, pl.wexp[pl.cur] + b, sizeof(int));
    const int rc = plan->be.sync();
    if (deg) *deg = pl.res_deg;
    if (W) *W = w;
    return rc;
}

// ---- section 2: private-layer seam, host buffers ----------------------------------------------

FNFT_UINT fnft__poly_fmult2x2_numel(const FNFT_UINT deg, const FNFT_UINT n)
{
    return 4 * (deg + 1) * (n == 0 ? 0 : nft_nextpow2(n));
}

FNFT_INT fnft__poly_fmult2x2(FNFT_UINT *const d, FNFT_UINT n, FNFT_COMPLEX *const p,
                             FNFT_COMPLEX *const result, FNFT_INT *const W_ptr)
{
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    HipBackend be;
    return api_poly_fmult2x2(be, d, n, p, result, W_ptr);
}

// fnft__poly_fmult2x2 on matrices that are already in device memory (the root's step of a sample-axis split: the G
// block matrices arrive by RCCL): d_p holds n matrices of degree deg in the reference's input layout, d_result
// (4 * (n*deg + 1) complex128) receives the product in the reference's result layout, normalised; *W_out its exponent.
// Runs on `stream` of the calling thread's current device; waits for the stream once (W is read back).
FNFT_INT fnft_amd_poly_fmult2x2_device(FNFT_UINT deg, FNFT_UINT n, const void *d_p, void *d_result, FNFT_UINT *deg_out,
                                       FNFT_INT *W_out, void *stream)
{
    if (deg == 0 || n == 0 || !d_p || !d_result || !W_out || deg > 0x7fffffff) return FNFT_EC_INVALID_ARGUMENT;
    if (current_device() < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    HipBackend be;
    be.stream = (hipStream_t)stream;
    Plan pl(be, n, 0, 1, -1, (int)deg);
    int rc = pl.init();
    if (rc == NFT_SUCCESS) rc = pl.load_level0_from_device(d_p);
    if (rc == NFT_SUCCESS) rc = pl.run_tree();
    if (rc == NFT_SUCCESS) {
        pl.export_tm((cplx *)d_result, (size_t)(pl.res_deg + 1), false);
        int W = 0;
        be.d2h(&W, pl.wexp[pl.cur], sizeof(int));
        rc = be.sync();
        *W_out = W;
        if (deg_out) *deg_out = pl.res_deg;
    }
    pl.destroy();
    if (be.failed) return FNFT_EC_OTHER;
    return rc;
}

// ---- scalar products and single pair products (src/private/fnft__poly_fmult.c:35-38,45-121,152-328) ----------
// They ride on the 2x2 tree: a scalar polynomial p is the matrix diag(p, p), whose products have the scalar
// product in entry 11.
FNFT_UINT fnft__poly_fmult_numel(const FNFT_UINT deg, const FNFT_UINT n)
{
    return (deg + 1) * (n == 0 ? 0 : nft_nextpow2(n));
}

FNFT_INT fnft__poly_fmult(FNFT_UINT *const d, FNFT_UINT n, FNFT_COMPLEX *const p, FNFT_INT *const W_ptr)
{
    if (!d || !p || n == 0 || *d == 0) return FNFT_EC_INVALID_ARGUMENT;
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    const size_t deg = *d, w = deg + 1;
    std::vector<std::complex<double>> m(fnft__poly_fmult2x2_numel(deg, n)), r(m.size());
    for (size_t i = 0; i < n * w; i++) {
        m[i] = p[i];                  // entry 11
        m[3 * n * w + i] = p[i];      // entry 22
    }
    HipBackend be;
    size_t dd = deg;
    const int rc = api_poly_fmult2x2(be, &dd, n, m.data(), r.data(), W_ptr);
    if (rc != FNFT_SUCCESS) return rc;
    for (size_t i = 0; i <= dd; i++) p[i] = r[i];
    *d = dd;
    return FNFT_SUCCESS;
}

// fft_wrapper_next_fft_length(2*(deg+1)-1), include/private/fnft__fft_wrapper.h:43-48 (kiss_fft_next_fast_size):
// the length callers of the reference size buf0..buf2 with
FNFT_INT fnft__poly_fmult_two_polys_len(const FNFT_UINT deg)
{
    size_t n = 2 * (deg + 1) - 1;
    for (;; n++) {
        size_t m = n;
        while (m % 2 == 0) m /= 2;
        while (m % 3 == 0) m /= 3;
        while (m % 5 == 0) m /= 5;
        if (m <= 1) return (FNFT_INT)n;
    }
}

static int two_polys_product(size_t deg, const std::complex<double> *p1, const std::complex<double> *p2,
                             std::complex<double> *out /* 2*deg+1 */)
{
    const size_t w = deg + 1;
    std::vector<std::complex<double>> m(fnft__poly_fmult2x2_numel(deg, 2)), r(m.size());
    for (size_t i = 0; i < w; i++) {
        m[i] = p1[i];           m[w + i] = p2[i];                  // entry 11 of both factors
        m[6 * w + i] = p1[i];   m[7 * w + i] = p2[i];              // entry 22
    }
    HipBackend be;
    size_t dd = deg;
    const int rc = api_poly_fmult2x2(be, &dd, 2, m.data(), r.data(), nullptr);
    if (rc != FNFT_SUCCESS) return rc;
    for (size_t i = 0; i <= 2 * deg; i++) out[i] = r[i];
    return FNFT_SUCCESS;
}

// src/private/fnft__poly_fmult.c:50-121.  plan_fwd / plan_inv are the reference's KissFFT / FFTW handles: the GPU
// does its own transforms and ignores them (pass whatever the caller has, or NULL).  The reference keeps the
// SPECTRA of p1 / p2 in buf1 / buf2 between calls (a NULL factor means "the one of the previous call") and, in
// modes 2 / 3, a partial sum of spectra in `result`; here the same buffers carry the COEFFICIENTS instead --
// what a sequence of calls returns in the coefficient domain (modes 0, 1, 3) is the same.
FNFT_INT fnft__poly_fmult_two_polys(const FNFT_UINT deg, FNFT_COMPLEX const *const p1, FNFT_COMPLEX const *const p2,
                                    FNFT_COMPLEX *const result, void *plan_fwd, void *plan_inv, FNFT_COMPLEX *const buf0,
                                    FNFT_COMPLEX *const buf1, FNFT_COMPLEX *const buf2, const FNFT_UINT mode)
{
    (void)plan_fwd; (void)plan_inv; (void)buf0;
    if (!result || !buf1 || !buf2 || mode > 3) return FNFT_EC_INVALID_ARGUMENT;
    if (current_device() < 0) return FNFT_EC_OTHER;
    const size_t w = deg + 1, wr = 2 * deg + 1;
    if (p1) for (size_t i = 0; i < w; i++) buf1[i] = p1[i];
    if (p2) for (size_t i = 0; i < w; i++) buf2[i] = p2[i];
    std::vector<std::complex<double>> prod(wr);
    if (deg == 0) prod[0] = buf1[0] * buf2[0];
    else {
        const int rc = two_polys_product(deg, buf1, buf2, prod.data());
        if (rc != FNFT_SUCCESS) return rc;
    }
    switch (mode) {
    case 0: for (size_t i = 0; i < wr; i++) result[i] = prod[i]; break;
    case 1: for (size_t i = 0; i < wr; i++) result[i] += prod[i]; break;
    case 2: for (size_t i = 0; i < wr; i++) result[i] = prod[i]; break;    // partial sum kept for a mode-3 call
    default: for (size_t i = 0; i < wr; i++) result[i] += prod[i]; break;  // mode 3: partial sum + this product
    }
    return FNFT_SUCCESS;
}

// src/private/fnft__poly_fmult.c:239-328: one 2x2 product, entries at the given strides; plans, buffers and
// mode_offset (where the reference parks spectra) are not needed and ignored
FNFT_INT fnft__poly_fmult_two_polys2x2(const FNFT_UINT deg, FNFT_COMPLEX const *const p1_11, const FNFT_UINT p1_stride,
                                       FNFT_COMPLEX const *const p2_11, const FNFT_UINT p2_stride,
                                       FNFT_COMPLEX *const result_11, const FNFT_UINT result_stride, void *plan_fwd,
                                       void *plan_inv, FNFT_COMPLEX *const buf0, FNFT_COMPLEX *const buf1,
                                       FNFT_COMPLEX *const buf2, const FNFT_UINT mode_offset)
{
    (void)plan_fwd; (void)plan_inv; (void)buf0; (void)buf1; (void)buf2; (void)mode_offset;
    if (!p1_11 || !p2_11 || !result_11 || deg == 0) return FNFT_EC_INVALID_ARGUMENT;
    if (current_device() < 0) return FNFT_EC_OTHER;
    const size_t w = deg + 1, wr = 2 * deg + 1;
    std::vector<std::complex<double>> m(fnft__poly_fmult2x2_numel(deg, 2)), r(m.size());
    for (int e = 0; e < 4; e++)
        for (size_t i = 0; i < w; i++) {
            m[(size_t)e * 2 * w + i] = p1_11[(size_t)e * p1_stride + i];
            m[(size_t)e * 2 * w + w + i] = p2_11[(size_t)e * p2_stride + i];
        }
    HipBackend be;
    size_t dd = deg;
    const int rc = api_poly_fmult2x2(be, &dd, 2, m.data(), r.data(), nullptr);
    if (rc != FNFT_SUCCESS) return rc;
    for (int e = 0; e < 4; e++)
        for (size_t i = 0; i < wr; i++) result_11[(size_t)e * result_stride + i] = r[(size_t)e * wr + i];
    return FNFT_SUCCESS;
}

// include/private/fnft__nse_finvscatter.h (src/private/fnft__nse_finvscatter.c:234-366): samples from a transfer
// matrix by layer peeling, every array on the device (NftLayerPeelingDev, nft_nsev_inverse.h)
// argument errors of the private seams are raised like the reference raises them (E_INVALID_ARGUMENT(name): code + text
// through the error hook, src/fnft_errwarn.c), not returned bare
extern "C" FNFT_INT fnft_amd__raise(FNFT_INT ec, const char *func, int line, const char *msg);   // fnft_nsev_host.c
static FNFT_INT seam_invalid(const char *func, int line, const char *arg)
{
    const std::string msg = std::string("Invalid argument ") + arg + ".";
    return fnft_amd__raise(FNFT_EC_INVALID_ARGUMENT, func, line, msg.c_str());
}
#define SEAM_CHECK(cond, arg) do { if (cond) return seam_invalid(__func__, __LINE__, #arg); } while (0)

FNFT_INT fnft__nse_finvscatter(const FNFT_UINT deg, FNFT_COMPLEX *const transfer_matrix, FNFT_COMPLEX *const q,
                               const FNFT_REAL eps_t, const FNFT_INT kappa, const fnft_nse_discretization_t discretization)
{
    // argument checks in the reference's order, :242-260
    SEAM_CHECK(deg == 0, deg);
    SEAM_CHECK(!transfer_matrix, transfer_matrix);
    SEAM_CHECK(!q, q);
    SEAM_CHECK(!(eps_t > 0.0), eps_t);
    SEAM_CHECK(kappa != -1 && kappa != 1, kappa);
    const int akns = nft_nse_to_akns((int)discretization);
    const int ddeg = akns < 0 ? 0 : nft_akns_degree(akns);
    SEAM_CHECK(ddeg == 0, discretization);
    const size_t D = deg / (size_t)ddeg;
    if (D < 2 || (D & (D - 1)) != 0) return FNFT_EC_OTHER;           // not a power of two, :259-260
    // the base case exists for the two degree-1 schemes only, :164-211
    const bool modal = discretization == fnft_nse_discretization_2SPLIT2_MODAL;
    SEAM_CHECK(!modal && discretization != fnft_nse_discretization_2SPLIT2A, discretization);
    if (current_device() < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    // one resident peeler per device: its pair-product plans (one per degree) and work arrays are reused by later
    // calls; sizes beyond 2^18 samples release everything again (workspace footprint)
    const int dev = current_device();
    Peeler *pp = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_peelers_mtx);
        Peeler *&slot = g_peelers[dev];
        if (!slot) slot = new (std::nothrow) Peeler();
        pp = slot;
    }
    if (!pp) return FNFT_EC_NOMEM;
    pp->be.failed = false;
    pp->lp.eps_t = eps_t;
    pp->lp.kappa = (int)kappa;
    pp->lp.modal = modal ? 1 : 0;
    const int rc = pp->lp.run_host(deg, transfer_matrix, q);
    const bool failed = pp->be.failed;
    if (failed || rc != FNFT_SUCCESS || deg > ((size_t)1 << 18)) pp->lp.destroy();
    if (failed) return FNFT_EC_OTHER;
    return rc;
}

// ---- fnft_nsev_inverse (driver: fnft_nsev_inverse_host.c) ---------------------------------------------------------
// include/private/fnft__poly_specfact.h (src/private/fnft__poly_specfact.c:25-140)
FNFT_INT fnft__poly_specfact(const FNFT_UINT deg, FNFT_COMPLEX const *const poly, FNFT_COMPLEX *const result,
                             const FNFT_UINT oversampling_factor, const FNFT_INT kappa)
{
    SEAM_CHECK(deg == 0, deg);                                   // :31-38
    SEAM_CHECK(!poly, poly);
    SEAM_CHECK(!result, result);
    SEAM_CHECK(oversampling_factor == 0, oversampling_factor);
    SEAM_CHECK(kappa != 0 && kappa != 1 && kappa != -1, kappa);  // :105-107
    if (current_device() < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    HipBackend be;
    NftInverseDev<HipBackend> inv(be);
    int warn = 0;
    const int rc = inv.specfact_host(deg, poly, result, oversampling_factor, (int)kappa, &warn);
    if (be.failed) return FNFT_EC_OTHER;
    if (rc == FNFT_SUCCESS && warn) fnft_amd__warn("Ill-posed spectral factorization problem.", "fnft__poly_specfact", __LINE__);
    return rc;
}

extern "C" FNFT_INT fnft_amd__inverse_transfer_matrix(FNFT_UINT M, FNFT_COMPLEX *contspec, const FNFT_REAL *XI, FNFT_UINT K,
                                                      const FNFT_COMPLEX *bound_states, FNFT_UINT D, const FNFT_REAL *T,
                                                      FNFT_UINT deg, FNFT_COMPLEX *tm, FNFT_INT kappa, int cstype,
                                                      int method, FNFT_UINT max_iter, FNFT_UINT oversampling,
                                                      FNFT_REAL phase_factor, int *warn_specfact, int *warn_maxiter)
{
    if (current_device() < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    HipBackend be;
    NftInverseDev<HipBackend> inv(be);
    const int rc = inv.transfer_matrix(M, contspec, XI, K, bound_states, D, T, deg, tm, (int)kappa, cstype, method,
                                       max_iter, oversampling, phase_factor, warn_specfact, warn_maxiter);
    if (be.failed) return FNFT_EC_OTHER;
    return rc;
}

// src/fnft_nsev_inverse.c:680-903: host bookkeeping (sorting, residues -> norming constants), device Darboux steps
extern "C" FNFT_INT fnft_amd__inverse_add_discrete(FNFT_UINT K, const FNFT_COMPLEX *bound_states,
                                                   const FNFT_COMPLEX *normconsts_or_residues, FNFT_UINT D,
                                                   FNFT_COMPLEX *q, const FNFT_REAL *T, int contspec_flag, int residues,
                                                   int seed_method)
{
    typedef std::complex<double> cd;
    if (K == 0 || !bound_states || !normconsts_or_residues) return FNFT_EC_INVALID_ARGUMENT;
    if (current_device() < 0) return FNFT_EC_OTHER;
    std::vector<cd> bs(bound_states, bound_states + K), nc(normconsts_or_residues, normconsts_or_residues + K);
    for (size_t i = 0; i < K; i++)            // descending imaginary part, the reference's exchange sort (:742-754)
        for (size_t j = i + 1; j < K; j++)
            if (bs[i].imag() < bs[j].imag()) { std::swap(bs[i], bs[j]); std::swap(nc[i], nc[j]); }
    for (size_t i = 0; i + 1 < K; i++)
        if (bs[i + 1] == bs[i]) return FNFT_EC_SANITY_CHECK_FAILED;   // :756-761
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    HipBackend be;
    int rc = FNFT_SUCCESS;
    if (residues) {                                                    // :771-795
        std::vector<cd> acs(K, cd(1.0, 0.0)), ap(K), bdummy(K);
        if (contspec_flag) {
            // the non-solitonic part of the potential contributes to the residues: a(lambda_k) of the seed, BO scheme
            NftDiscSpec<HipBackend> ds(be);
            NftDiscSpec<HipBackend>::Prepared P;
            rc = ds.prepare(D, q, T, D, (int)fnft_nse_discretization_2SPLIT4B, P, false);
            if (rc == FNFT_SUCCESS) rc = ds.scatter(P, K, bs.data(), acs.data(), ap.data(), bdummy.data(), true);
            ds.release(P);
            if (rc != FNFT_SUCCESS || be.failed) return be.failed ? FNFT_EC_OTHER : rc;
        }
        for (size_t i = 0; i < K; i++) {
            cd tmp = acs[i];
            for (size_t j = 0; j < K; j++)
                if (j != i) tmp = tmp * (bs[i] - bs[j]) / (bs[i] - std::conj(bs[j]));
            nc[i] = (nc[i] / cd(0.0, 2.0 * bs[i].imag())) * tmp;
        }
    }
    int mode;
    if (contspec_flag == 0 && !seed_method) mode = 0;
    else if ((contspec_flag == 0 && seed_method) || (contspec_flag == 1 && !seed_method)) mode = 1;
    else return FNFT_EC_INVALID_ARGUMENT;                              // :890-891
    NftInverseDev<HipBackend> inv(be);
    rc = inv.add_discrete(K, bs.data(), nc.data(), D, q, T, mode);
    if (be.failed) return FNFT_EC_OTHER;
    return rc;
}

FNFT_INT fnft_amd_poly_chirpz(const FNFT_UINT deg, FNFT_COMPLEX const *const p, const double *A,
                              const double *W, const FNFT_UINT M, FNFT_COMPLEX *const result)
{
    SEAM_CHECK(!p, p);
    SEAM_CHECK(M == 0, M);
    SEAM_CHECK(!result, result);
    SEAM_CHECK(!A || !W, A);
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    HipBackend be;
    return Plan::chirpz_host(be, deg, p, {A[0], A[1]}, {W[0], W[1]}, M, result);
}

FNFT_INT fnft__poly_chirpz(const FNFT_UINT deg, FNFT_COMPLEX const *const p, const FNFT_COMPLEX A,
                           const FNFT_COMPLEX W, const FNFT_UINT M, FNFT_COMPLEX *const result)
{
    const double a[2] = {A.real(), A.imag()}, w[2] = {W.real(), W.imag()};
    return fnft_amd_poly_chirpz(deg, p, a, w, M, result);
}

// include/private/fnft__misc.h:241-242 (src/private/fnft__misc.c:326-407): band-limited shift by delta
FNFT_INT fnft__misc_resample(const FNFT_UINT D, const FNFT_REAL eps_t, FNFT_COMPLEX const *const q,
                             const FNFT_REAL delta, FNFT_COMPLEX *const q_new)
{
    // argument checks of the reference, :331-338
    SEAM_CHECK(D <= 2, D);
    SEAM_CHECK(!q, q);
    SEAM_CHECK(!q_new, q_new);
    SEAM_CHECK(eps_t == 0.0, eps_t);
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    HipBackend be;
    int warn = 0;
    const int rc = Plan::resample_host(be, D, eps_t, q, delta, q_new, &warn);
    if (rc == FNFT_SUCCESS && warn) fnft_amd__warn(kNotBandlimitedMsg, "fnft__misc_resample", __LINE__);
    return rc;
}

// include/private/fnft__poly_roots_fasteigen.h (src/private/fnft__poly_roots_fasteigen.c:29-48): all roots of
// p[0] z^deg + ... + p[deg].  The reference calls eiscor's QR (Fortran); here the Ehrlich-Aberth kernels.
FNFT_INT fnft__poly_roots_fasteigen(const FNFT_UINT deg, FNFT_COMPLEX const *const p, FNFT_COMPLEX *const roots)
{
    SEAM_CHECK(!p, p);
    SEAM_CHECK(!roots, roots);
    if (deg == 0) return FNFT_SUCCESS;
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    HipBackend be;
    NftDiscSpec<HipBackend> ds(be);
    cplx *d_coef = (cplx *)be.alloc((deg + 1) * sizeof(cplx));
    if (!d_coef) return FNFT_EC_NOMEM;
    be.h2d(d_coef, p, (deg + 1) * sizeof(cplx));
    std::vector<std::complex<double>> z;
    int rc = ds.roots(d_coef, deg, z);
    be.free(d_coef);
    if (be.failed) return FNFT_EC_OTHER;
    if (rc != NFT_SUCCESS) return -abs(rc);   // E_SUBROUTINE, :44-47
    for (size_t i = 0; i < deg; i++) roots[i] = z[i];
    return FNFT_SUCCESS;
}

// include/private/fnft__nse_scatter.h:76-80 (src/private/fnft__nse_scatter_bound_states.c:29-667) for the
// Boffetta-Osborne scheme (the one fnft_nsev uses for Newton refinement and norming constants of every
// 2SPLIT discretization): a, a' and b = phi/psi at K points lambda; q holds the D samples, r is not read
// (r = -conj(q)).  Chunk-parallel on the GPU (body_bs_*).
FNFT_INT fnft__nse_scatter_bound_states(const FNFT_UINT D, FNFT_COMPLEX const *const q, FNFT_COMPLEX *r,
                                        FNFT_REAL const *const T, FNFT_UINT K, FNFT_COMPLEX *bound_states,
                                        FNFT_COMPLEX *a_vals, FNFT_COMPLEX *aprime_vals, FNFT_COMPLEX *b,
                                        fnft_nse_discretization_t discretization, FNFT_UINT skip_b_flag)
{
    (void)r;
    // argument checks in the reference's order, :53-66
    SEAM_CHECK(D == 0, D);
    SEAM_CHECK(!q, q);
    SEAM_CHECK(!T, eps_t);
    SEAM_CHECK(K == 0, K);
    SEAM_CHECK(!bound_states, bound_states);
    SEAM_CHECK(!a_vals, a);
    SEAM_CHECK(!aprime_vals, a_prime);
    SEAM_CHECK(!skip_b_flag && !b, b);
    const bool cf42 = discretization == fnft_nse_discretization_CF4_2;
    if (discretization != fnft_nse_discretization_BO && !cf42) return FNFT_EC_NOT_YET_IMPLEMENTED;
    if (D < 2 || !(T[0] < T[1])) return FNFT_EC_INVALID_ARGUMENT;
    if (cf42 && (D % 2 != 0 || D < 4)) return FNFT_EC_ASSERTION_FAILED;   // :188-191
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    HipBackend be;
    NftDiscSpec<HipBackend> ds(be);
    NftDiscSpec<HipBackend>::Prepared P;
    int rc;
    if (cf42) {
        // CF4_2 (the scatterer fnft_nsev uses with 4SPLIT4A/B, :188-197): q holds the D preprocessed samples, two per
        // grid point; the grid has D/2 points on T
        P.ups = 2; P.deg0 = 4;
        P.Deff = D; P.Dsub = D / 2;
        P.T[0] = T[0]; P.T[1] = T[1];
        P.eps_t = (T[1] - T[0]) / (double)(D / 2 - 1);
        P.d_in = (cplx *)be.alloc(D * sizeof(cplx));
        rc = P.d_in ? NFT_SUCCESS : NFT_EC_NOMEM;
        if (rc == NFT_SUCCESS) {
            be.h2d(P.d_in, q, D * sizeof(cplx));
            P.d_qpre = P.d_in;
        }
    } else {
        // any 2SPLIT scheme: upsampling factor 1, the slow scatterer is BO (src/fnft_nsev.c:675-680)
        rc = ds.prepare(D, (const std::complex<double> *)q, T, D, (int)fnft_nse_discretization_2SPLIT4B, P, false);
    }
    if (rc == NFT_SUCCESS)
        rc = ds.scatter(P, K, (const std::complex<double> *)bound_states, (std::complex<double> *)a_vals,
                        (std::complex<double> *)aprime_vals, (std::complex<double> *)b, skip_b_flag != 0);
    ds.release(P);
    if (be.failed) return FNFT_EC_OTHER;
    return rc;
}

// include/private/fnft__poly_roots_fftgridsearch.h (src/private/fnft__poly_roots_fftgridsearch.c:35-151 and :159-217):
// roots of a polynomial on the arc PHI of the unit circle.  *M_ptr: grid points in, estimates out; roots: *M_ptr
// entries.  The chirp z-transforms, the candidate test and the ordered compaction run on the GPU.
static int gridsearch_common(const size_t deg, const std::complex<double> *p, size_t *M_ptr, const double *PHI,
                             std::complex<double> *roots, bool paraherm)
{
    const size_t M = *M_ptr;
    const double eps = (PHI[1] - PHI[0]) / (double)(M - 1);
    const std::complex<double> W(std::cos(eps), std::sin(eps));
    HipBackend be;
    const size_t nblk = (M + 255) / 256, rings = paraherm ? 1 : 3;
    cplx *vals = (cplx *)be.alloc(rings * M * sizeof(cplx)), *cand = (cplx *)be.alloc(M * sizeof(cplx));
    cplx *out = (cplx *)be.alloc(M * sizeof(cplx));
    int *keep = (int *)be.alloc(M * sizeof(int)), *bcnt = (int *)be.alloc(nblk * sizeof(int));
    int *boff = (int *)be.alloc(nblk * sizeof(int)), *dstatus = (int *)be.alloc(4 * sizeof(int));
    int rc = (vals && cand && out && keep && bcnt && boff && dstatus) ? FNFT_SUCCESS : FNFT_EC_NOMEM;
    size_t nroots = 0;
    if (rc == FNFT_SUCCESS) {
        be.memset0(dstatus, 4 * sizeof(int));
        for (size_t k = 0; k < rings && rc == FNFT_SUCCESS; k++) {
            const double scl = paraherm ? 1.0 : 1.0 + ((double)k - 1.0) * eps;      // rings k = -1, 0, 1 (:66-72)
            const std::complex<double> A = scl * std::complex<double>(std::cos(PHI[0]), -std::sin(PHI[0]));
            rc = Plan::chirpz_host(be, deg, p, A, W, M, nullptr, vals + k * M);
        }
    }
    if (rc == FNFT_SUCCESS) {
        GridSearchParams G;
        std::memset(&G, 0, sizeof(G));
        G.vals = vals; G.M = (long long)M; G.phi0 = PHI[0]; G.eps = eps; G.N1 = (long long)(deg / 2);
        G.keep = keep; G.cand = cand; G.blockcnt = bcnt; G.blockoff = boff; G.out = out; G.status = dstatus;
        if (paraherm) be.run<KGridMarkPh>((int)nblk, 1, G);
        else be.run<KGridMark>((int)nblk, 1, G);
        be.run<KCompactCount>((int)nblk, 1, G);
        std::vector<int> hc(nblk), ho(nblk);
        int hst[4] = {0, 0, 0, 0};
        be.d2h(hc.data(), bcnt, nblk * sizeof(int));
        be.d2h(hst, dstatus, sizeof(hst));
        rc = be.sync();
        if (rc == FNFT_SUCCESS && (hst[0] & 2)) rc = FNFT_EC_DIV_BY_ZERO;
        if (rc == FNFT_SUCCESS) {
            for (size_t b = 0; b < nblk; b++) { ho[b] = (int)nroots; nroots += (size_t)hc[b]; }
            be.h2d(boff, ho.data(), nblk * sizeof(int));
            be.run<KCompactScatter>((int)nblk, 1, G);
            if (nroots) be.d2h(roots, out, nroots * sizeof(cplx));
            rc = be.sync();
        }
    }
    be.free(vals); be.free(cand); be.free(out); be.free(keep); be.free(bcnt); be.free(boff); be.free(dstatus);
    if (be.failed) return FNFT_EC_OTHER;
    if (rc == FNFT_SUCCESS) *M_ptr = nroots;
    return rc;
}
FNFT_INT fnft__poly_roots_fftgridsearch(const FNFT_UINT deg, FNFT_COMPLEX const *const p, FNFT_UINT *const M_ptr,
                                        FNFT_REAL const *const PHI, FNFT_COMPLEX *const roots)
{
    // argument checks in the reference's order, :46-57
    SEAM_CHECK(deg < 2, deg);
    SEAM_CHECK(!p, p);
    SEAM_CHECK(!M_ptr || *M_ptr < 2, M_ptr);
    SEAM_CHECK(!PHI || !(PHI[0] < PHI[1]) || PHI[0] == -INFINITY || PHI[1] == INFINITY, PHI);
    SEAM_CHECK(!roots, roots);
    if (current_device() < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    size_t M = *M_ptr;
    const int rc = gridsearch_common(deg, p, &M, PHI, roots, false);
    if (rc == FNFT_SUCCESS) *M_ptr = M;
    return rc;
}
FNFT_INT fnft__poly_roots_fftgridsearch_paraherm(const FNFT_UINT deg, FNFT_COMPLEX const *const p, FNFT_UINT *const M_ptr,
                                                 FNFT_REAL const *const PHI, FNFT_COMPLEX *const roots)
{
    // :170-180: the degree must be even
    SEAM_CHECK(deg % 2 == 1 || deg < 2, deg);
    SEAM_CHECK(!p, p);
    SEAM_CHECK(!M_ptr || *M_ptr < 2, M_ptr);
    SEAM_CHECK(!PHI || !(PHI[0] < PHI[1]) || PHI[0] == -INFINITY || PHI[1] == INFINITY, PHI);
    SEAM_CHECK(!roots, roots);
    if (current_device() < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    size_t M = *M_ptr;
    const int rc = gridsearch_common(deg, p, &M, PHI, roots, true);
    if (rc == FNFT_SUCCESS) *M_ptr = M;
    return rc;
}

// include/private/fnft__nse_scatter.h:119-123 (src/private/fnft__nse_scatter_matrix.c:33-86 ->
// fnft__akns_scatter_matrix.c, BO scheme): S(lambda) = U_{D-1} ... U_0 and dS/dlambda for K values of lambda -- the slow
// scatterer the periodic problem and Newton refinements call.  Chunk-parallel like the bound-state scatterer.
FNFT_INT fnft__nse_scatter_matrix(const FNFT_UINT D, FNFT_COMPLEX const *const q, FNFT_COMPLEX *r, const FNFT_REAL eps_t,
                                  const FNFT_INT kappa, const FNFT_UINT K, FNFT_COMPLEX const *const lambda,
                                  FNFT_COMPLEX *const result, fnft_nse_discretization_t discretization,
                                  const FNFT_UINT derivative_flag)
{
    // argument checks in the reference's order, :46-59
    if (D == 0 || !q || !(eps_t > 0) || (kappa != 1 && kappa != -1) || K == 0 || !lambda || !result)
        return FNFT_EC_INVALID_ARGUMENT;
    // BO, and CF4_2 (two preprocessed samples per step, each advanced with lambda/2 and the whole step eps_t; the
    // derivative picks up the factor 1/2 -- BsParams::lscale, applied by the chunk kernel:
    // src/private/fnft__akns_scatter_matrix.c:122-130,201-208)
    const bool cf42 = discretization == fnft_nse_discretization_CF4_2;
    if (discretization != fnft_nse_discretization_BO && !cf42) return FNFT_EC_NOT_YET_IMPLEMENTED;
    if (cf42 && D % 2 != 0) return FNFT_EC_ASSERTION_FAILED;
    if (current_device() < 0) return FNFT_EC_OTHER;
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    HipBackend be;
    BsParams B;
    std::memset(&B, 0, sizeof(B));
    size_t L = (D + 16383) / 16384;
    if (L < 16) L = 16;
    if (L % 2) L++;
    const size_t nchunk = (D + L - 1) / L;
    const size_t w = derivative_flag ? 8 : 4;
    cplx *dq = (cplx *)be.alloc(D * sizeof(cplx)), *dr = r ? (cplx *)be.alloc(D * sizeof(cplx)) : nullptr;
    cplx *dl = (cplx *)be.alloc(K * sizeof(cplx)), *cm = (cplx *)be.alloc(K * nchunk * 8 * sizeof(cplx));
    cplx *ds = (cplx *)be.alloc(K * w * sizeof(cplx));
    int rc = (dq && dl && cm && ds && (!r || dr)) ? FNFT_SUCCESS : FNFT_EC_NOMEM;
    if (rc == FNFT_SUCCESS) {
        be.h2d(dq, q, D * sizeof(cplx));
        if (r) be.h2d(dr, r, D * sizeof(cplx));
        be.h2d(dl, lambda, K * sizeof(cplx));
        B.q = dq; B.r = dr; B.kappa = (int)kappa; B.D = (long long)D; B.ups = cf42 ? 2 : 1; B.lscale = cf42 ? 0.5 : 1.0;
        B.eps = eps_t;
        B.K = (int)K; B.lam = dl; B.L = (int)L; B.nchunk = (int)nchunk; B.cm = cm; B.smat = ds;
        B.with_deriv = derivative_flag ? 1 : 0;
        for (size_t k0 = 0; k0 < K; k0 += 32768) {   // grid.y limit
            const size_t kn = std::min<size_t>(32768, K - k0);
            BsParams Bk = B;
            Bk.K = (int)kn; Bk.lam = dl + k0; Bk.cm = cm + k0 * nchunk * 8; Bk.smat = ds + k0 * w;
            be.run<KBsChunk<false>>((int)((nchunk + 63) / 64), (int)kn, Bk);
            be.run<KBsMatrix>((int)kn, 1, Bk);
        }
        be.d2h(result, ds, K * w * sizeof(cplx));
        rc = be.sync();
    }
    be.free(dq); be.free(dr); be.free(dl); be.free(cm); be.free(ds);
    if (be.failed) return FNFT_EC_OTHER;
    return rc;
}

FNFT_UINT fnft__akns_fscatter_numel(FNFT_UINT D, fnft__akns_discretization_t discretization)
{
    const int deg = nft_akns_degree((int)discretization);
    return deg == 0 ? 0 : fnft__poly_fmult2x2_numel((FNFT_UINT)deg, D);
}

FNFT_INT fnft__akns_fscatter(const FNFT_UINT D, FNFT_COMPLEX const *const q,
                             FNFT_COMPLEX const *const r, const FNFT_REAL eps_t,
                             FNFT_COMPLEX *const result, FNFT_UINT *const deg_ptr,
                             FNFT_INT *const W_ptr, fnft__akns_discretization_t discretization)
{
    // argument checks in the reference's order, src/private/fnft__akns_fscatter.c:80-97
    SEAM_CHECK(D == 0, D);
    SEAM_CHECK(!q, q);
    SEAM_CHECK(!r, r);
    SEAM_CHECK(!(eps_t > 0.0), eps_t);
    SEAM_CHECK(!result, result);
    SEAM_CHECK(!deg_ptr, deg_ptr);
    SEAM_CHECK(nft_akns_degree((int)discretization) == 0, discretization);
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    HipBackend be;
    return api_akns_fscatter(be, D, q, r, eps_t, 1, result, deg_ptr, W_ptr, (int)discretization);
}

FNFT_UINT fnft__nse_fscatter_numel(FNFT_UINT D, fnft_nse_discretization_t discretization)
{
    const int a = nft_nse_to_akns((int)discretization);
    return a < 0 ? 0 : fnft__akns_fscatter_numel(D, (fnft__akns_discretization_t)a);
}

FNFT_INT fnft__nse_fscatter(const FNFT_UINT D, FNFT_COMPLEX const *const q, const FNFT_REAL eps_t,
                            const FNFT_INT kappa, FNFT_COMPLEX *const result,
                            FNFT_UINT *const deg_ptr, FNFT_INT *const W_ptr,
                            fnft_nse_discretization_t discretization)
{
    // src/private/fnft__nse_fscatter.c:55-69
    if (D == 0 || !q || !(eps_t > 0.0) || (kappa != 1 && kappa != -1) || !result || !deg_ptr)
        return FNFT_EC_INVALID_ARGUMENT;
    const int a = nft_nse_to_akns((int)discretization);
    if (a < 0) return FNFT_EC_INVALID_ARGUMENT;
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    HipBackend be;
    return api_akns_fscatter(be, D, q, nullptr, eps_t, kappa, result, deg_ptr, W_ptr, a);
}

// 4SPLIT4A / 4SPLIT4B of the KdV seams: per-sample formulas and degrees of 2SPLIT4A / 2SPLIT4B on the samples as they
// are (src/private/fnft__kdv_discretization.c:139-143; kdv_fscatter does not resample)
static int kdv_alias(int kd)
{
    if (kd == (int)fnft_kdv_discretization_4SPLIT4A) return (int)fnft_kdv_discretization_2SPLIT4A;
    if (kd == (int)fnft_kdv_discretization_4SPLIT4B) return (int)fnft_kdv_discretization_2SPLIT4B;
    return kd;
}

FNFT_UINT fnft__kdv_fscatter_numel(FNFT_UINT D, fnft_kdv_discretization_t discretization)
{
    const int kd = kdv_alias((int)discretization);
    if (kd < 0 || kd > (int)fnft_kdv_discretization_2SPLIT8B) return 0;
    return fnft__poly_fmult2x2_numel((FNFT_UINT)nft_akns_degree(kd + 1), D);
}

// src/private/fnft__kdv_fscatter.c:45-83
FNFT_INT fnft__kdv_fscatter(const FNFT_UINT D, FNFT_COMPLEX const *const u, const FNFT_REAL eps_t,
                            FNFT_COMPLEX *const result, FNFT_UINT *const deg_ptr, FNFT_INT *const W_ptr,
                            fnft_kdv_discretization_t discretization)
{
    SEAM_CHECK(D == 0, D);
    SEAM_CHECK(!u, u);
    SEAM_CHECK(!(eps_t > 0.0), eps_t);
    SEAM_CHECK(!result, result);
    SEAM_CHECK(!deg_ptr, deg_ptr);
    const int kd = kdv_alias((int)discretization);
    if (kd < 0 || kd > (int)fnft_kdv_discretization_2SPLIT8B) return FNFT_EC_INVALID_ARGUMENT;
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    HipBackend be;
    return api_kdv_fscatter(be, D, (const std::complex<double> *)u, eps_t, (std::complex<double> *)result, deg_ptr,
                            W_ptr, kd + 1);
}

// internal entry used by fnft_kdvv_host.c: host buffers in and out, plans cached per (D, M, scheme)
FNFT_INT fnft_amd__kdvv_contspec_host(FNFT_UINT D, const FNFT_COMPLEX *u, const FNFT_REAL *T, FNFT_UINT M,
                                      FNFT_COMPLEX *contspec, const FNFT_REAL *XI, int discretization)
{
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    fnft_amd_plan *P = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mtx);
        auto &cache = g_kdvv_cache;
        auto key = std::make_tuple(dev, (size_t)D, (size_t)M, discretization);
        auto it = cache.find(key);
        if (it == cache.end()) {
            if (cache.size() >= 4) {
                for (auto &kv : cache) fnft_amd_plan_destroy(kv.second);
                cache.clear();
            }
            const FNFT_INT rc = fnft_amd_kdvv_plan_create(&P, D, M, 1, (fnft_kdv_discretization_t)discretization, dev);
            if (rc != FNFT_SUCCESS) return rc;
            cache[key] = P;
        } else {
            P = it->second;
        }
    }
    // staging buffers on the device: from the block cache (no hipMalloc / hipFree per call)
    cplx *du = (cplx *)fa_pool_alloc(D * sizeof(cplx)), *dcs = (cplx *)fa_pool_alloc(M * sizeof(cplx));
    if (!du || !dcs) {
        if (du) fa_pool_free(du);
        if (dcs) fa_pool_free(dcs);
        return FNFT_EC_NOMEM;
    }
    {   // the potential is in host memory: look at it here instead of asking the device
        bool real = true;
        const std::complex<double> *uh = (const std::complex<double> *)u;
        for (size_t i = 0; i < D && real; i++) real = (uh[i].imag() == 0.0);
        P->real_mode = real ? 1 : 0;
    }
    FNFT_INT rc = FNFT_SUCCESS;
    if (!hip_ok(hipMemcpy(du, u, D * sizeof(cplx), hipMemcpyHostToDevice), "hipMemcpy(H2D)")) rc = FNFT_EC_OTHER;
    if (rc == FNFT_SUCCESS) rc = fnft_amd_kdvv_contspec_device(P, du, dcs, T, XI, nullptr);
    if (rc == FNFT_SUCCESS) rc = fnft_amd_plan_finish(P, nullptr);
    if (rc == FNFT_SUCCESS)
        if (!hip_ok(hipMemcpy(contspec, dcs, M * sizeof(cplx), hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
            rc = FNFT_EC_OTHER;
    fa_pool_free(du);
    fa_pool_free(dcs);
    return rc;
}

// internal entry used by fnft_nsev_host.c: discrete spectrum (kappa = +1), host buffers.
// *K_ptr: capacity of bound_states in, number of bound states out.  *warn: more were found than fit.
FNFT_INT fnft_amd__nsev_discspec_host(FNFT_UINT D, const FNFT_COMPLEX *q, const FNFT_REAL *T, int bsfilt,
                                      int bsloc, FNFT_UINT niter, FNFT_UINT Dsub, int dstype, int discretization,
                                      int richardson, FNFT_UINT *K_ptr, FNFT_COMPLEX *bound_states,
                                      FNFT_COMPLEX *normconsts_or_residues, int *warn)
{
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    HipBackend be;
    NftDiscSpec<HipBackend> ds(be);
    NftDsOpts o;
    o.bsfilt = bsfilt; o.bsloc = bsloc; o.niter = niter; o.Dsub = Dsub; o.dstype = dstype;
    o.nse_disc = discretization; o.richardson = richardson;
    size_t K = *K_ptr;
    const int rc = ds.run(D, (const std::complex<double> *)q, T, o, &K, (std::complex<double> *)bound_states,
                          (std::complex<double> *)normconsts_or_residues);
    if (warn) *warn = (ds.warn_more_than_K ? 1 : 0) | (ds.warn_roots_unconverged ? 2 : 0);
    if (be.failed) return FNFT_EC_OTHER;
    if (rc == NFT_SUCCESS) *K_ptr = K;
    return rc;
}

// ---- internal entry used by the C driver (fnft_nsev_host.c) ------------------------------------
// Host buffers in, host buffers out; plans are cached per (D, M, discretization, nskip).
// nskip > 1 (4SPLIT4A/B only): the transform of every nskip-th step of the D samples, as
// fnft__nse_discretization_preprocess_signal forms it for Richardson extrapolation.
FNFT_INT fnft_amd__nsev_contspec_host(FNFT_UINT D, const FNFT_COMPLEX *q, const FNFT_REAL *T,
                                      FNFT_UINT M, FNFT_COMPLEX *contspec, const FNFT_REAL *XI,
                                      FNFT_INT kappa, int discretization, int contspec_type,
                                      FNFT_INT normalization_flag, FNFT_UINT nskip)
{
    std::lock_guard<std::mutex> host_lk(host_call_mutex());
    const int dev = current_device();
    if (dev < 0) return FNFT_EC_OTHER;
    fnft_amd_plan *P = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_cache_mtx);
        auto &cache = g_nsev_cache;
        auto key = std::make_tuple(dev, (size_t)D, (size_t)M, discretization, (size_t)nskip);
        auto it = cache.find(key);
        if (it == cache.end()) {
            if (cache.size() >= 4) {  // keep the workspace footprint bounded
                for (auto &kv : cache) fnft_amd_plan_destroy(kv.second);
                cache.clear();
            }
            const FNFT_INT rc = fnft_amd_plan_create_sub(&P, D, M, 1,
                                                         (fnft_nse_discretization_t)discretization, dev, nskip);
            if (rc != FNFT_SUCCESS) return rc;
            cache[key] = P;
        } else {
            P = it->second;
        }
    }
    const size_t cs_len = M * (contspec_type == 0 ? 1 : (contspec_type == 1 ? 2 : 3));
    // staging buffers on the device: from the block cache (no hipMalloc / hipFree per call).  The caller's arrays are
    // ordinary pageable memory: hipMemcpy moves them at the link rate once their pages are resident.
    cplx *dq = (cplx *)fa_pool_alloc(D * sizeof(cplx));
    cplx *dcs = (cplx *)fa_pool_alloc((cs_len ? cs_len : 1) * sizeof(cplx));
    if (!dq || !dcs) {
        if (dq) fa_pool_free(dq);
        if (dcs) fa_pool_free(dcs);
        return FNFT_EC_NOMEM;
    }
    FNFT_INT rc = FNFT_SUCCESS;
    if (!hip_ok(hipMemcpy(dq, q, D * sizeof(cplx), hipMemcpyHostToDevice), "hipMemcpy(H2D)"))
        rc = FNFT_EC_OTHER;
    if (rc == FNFT_SUCCESS)
        rc = fnft_amd_nsev_contspec_device(P, dq, contspec ? dcs : nullptr, T, XI, kappa,
                                           (fnft_nsev_cstype_t)contspec_type, normalization_flag,
                                           nullptr);
    if (rc == FNFT_SUCCESS) rc = fnft_amd_plan_finish(P, nullptr);
    // the resampler of the 4SPLIT4A/B front end warns when the spectrum has not decayed (fnft__misc.c:371-381)
    if (fnft_amd_plan_last_warnings(P) & 1) fnft_amd__warn(kNotBandlimitedMsg, "fnft__misc_resample", __LINE__);
    if (rc == FNFT_SUCCESS && contspec && cs_len)
        if (!hip_ok(hipMemcpy(contspec, dcs, cs_len * sizeof(cplx), hipMemcpyDeviceToHost), "hipMemcpy(D2H)"))
            rc = FNFT_EC_OTHER;
    fa_pool_free(dq);
    fa_pool_free(dcs);
    return rc;
}

// Everything the host-pointer entry points keep on `device` between calls goes back to the driver: cached plans, the
// resident layer-peeling state of fnft__nse_finvscatter / fnft_nsev_inverse, and the cache of released device blocks.
// device < 0: every device.  Device-resident plans the caller created are not touched.
void fnft_amd_release_cached(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return;
    for (int d = 0; d < ndev; d++) {
        if (device >= 0 && d != device) continue;
        std::lock_guard<std::mutex> host_lk(host_call_mutex_of(d));
        {
            std::lock_guard<std::mutex> lk(g_cache_mtx);
            for (auto it = g_nsev_cache.begin(); it != g_nsev_cache.end();) {
                if (std::get<0>(it->first) == d) { fnft_amd_plan_destroy(it->second); it = g_nsev_cache.erase(it); }
                else ++it;
            }
            for (auto it = g_kdvv_cache.begin(); it != g_kdvv_cache.end();) {
                if (std::get<0>(it->first) == d) { fnft_amd_plan_destroy(it->second); it = g_kdvv_cache.erase(it); }
                else ++it;
            }
        }
        {
            std::lock_guard<std::mutex> lk(g_peelers_mtx);
            auto it = g_peelers.find(d);
            if (it != g_peelers.end() && it->second) {
                DeviceGuard dg(d);
                if (dg.ok) {
                    (void)hipDeviceSynchronize();
                    it->second->lp.destroy();
                }
            }
        }
        DeviceGuard dg(d);
        if (dg.ok) (void)hipDeviceSynchronize();
        DevPool &P = pool();
        std::lock_guard<std::mutex> lk(P.m);
        P.trim_locked(d);
    }
}

}  // extern "C"
