// ffk_api_resident.hip -- resident results: the user-facing PulseSequence.get_filter_function /
// ff.infidelity call with one H2D, one pass of ffk_pipeline_dev and one D2H of the small results
// (R stays in HBM), concatenations that read resident control matrices in place, hipGraph replay
// of repeated passes; and the host-logic self test of the sanitizer build.
#include <thread>

#include "ffk_api_common.h"


namespace {

struct Block {
    void* ptr;
    size_t size;
    int device = -1;      // device blocks belong to one device; pinned host blocks are portable
};

// Grow-only pools of device and pinned-host blocks: a PulseSequence is short-lived in user code
// (one per gate), hipMalloc / hipHostMalloc cost more than the whole pass at config 2.
struct BlockPool {
    std::mutex mu;
    std::vector<Block> free_blocks;
    bool pinned;
    explicit BlockPool(bool p) : pinned(p) {}
    int take(size_t bytes, int device, Block* out) {
        std::lock_guard<std::mutex> lock(mu);
        int best = -1;
        for (int i = 0; i < int(free_blocks.size()); ++i)
            if (free_blocks[i].size >= bytes && free_blocks[i].size <= 2*bytes + (1 << 16) &&
                (pinned || free_blocks[i].device == device) &&
                (best < 0 || free_blocks[i].size < free_blocks[best].size))
                best = i;
        if (best >= 0) {
            *out = free_blocks[best];
            free_blocks.erase(free_blocks.begin() + best);
            return FFK_OK;
        }
        const size_t want = align_up(bytes, size_t(1) << 16);
        void* p = nullptr;
        if (pinned)
            FFK_HIP(hipHostMalloc(&p, want, hipHostMallocPortable));
        else
            FFK_HIP(hipMalloc(&p, want));
        *out = {p, want, pinned ? -1 : device};
        return FFK_OK;
    }
    void give(Block b) {
        if (!b.ptr) return;
        std::lock_guard<std::mutex> lock(mu);
        if (free_blocks.size() >= 16) {       // bound what an idle process keeps: the OLDEST idle
            const Block old = free_blocks.front();   // block goes (a loop over fresh pulses of one
            free_blocks.erase(free_blocks.begin());  // shape must find its block again even after
            if (pinned) (void)hipHostFree(old.ptr); else (void)hipFree(old.ptr);   // other shapes filled the pool)
        }
        free_blocks.push_back(b);
    }
    int release() {
        std::lock_guard<std::mutex> lock(mu);
        for (Block& b : free_blocks) {
            if (pinned) FFK_HIP(hipHostFree(b.ptr)); else FFK_HIP(hipFree(b.ptr));
        }
        free_blocks.clear();
        return FFK_OK;
    }
};
BlockPool g_dev_pool(false), g_pin_pool(true);
// one stream per device for the resident passes, created on first use
std::mutex g_resident_stream_mu;
hipStream_t g_resident_streams[64] = {};

int resident_stream(hipStream_t* out) {
    int dev = 0;
    FFK_HIP(hipGetDevice(&dev));
    FFK_REQUIRE(dev >= 0 && dev < 64, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_resident_stream_mu);
    if (!g_resident_streams[dev])
        FFK_HIP(hipStreamCreateWithFlags(&g_resident_streams[dev], hipStreamNonBlocking));
    *out = g_resident_streams[dev];
    return FFK_OK;
}

// byte offsets of the arrays inside the device block and (first two groups) the pinned block
struct ResidentLayout {
    size_t H, dt, t, omega, basis, n_opers, n_coeffs, inputs_end;     // one H2D
    size_t D, V, Q, F, status, outputs_end;                            // one D2H
    size_t R, S, idx, infid, end;                                      // device only (+ infid D2H)
};
ResidentLayout resident_layout(int G, int d, int W, int N, int A) {
    ResidentLayout L;
    const size_t dd = size_t(d)*d;
    size_t o = 0;
    auto put = [&o](size_t bytes) { const size_t at = o; o += align_up(bytes); return at; };
    L.H = put(16*size_t(G)*dd);
    L.dt = put(8*size_t(G));
    L.t = put(8*size_t(G + 1));
    L.omega = put(8*size_t(W));
    L.basis = put(16*size_t(N)*dd);
    L.n_opers = put(16*size_t(A)*dd);
    L.n_coeffs = put(8*size_t(A)*G);
    L.inputs_end = o;
    L.D = put(8*size_t(G)*d);
    L.V = put(16*size_t(G)*dd);
    L.Q = put(16*size_t(G + 1)*dd);
    L.F = put(16*size_t(A)*A*W);
    L.status = put(sizeof(int32_t));
    L.outputs_end = o;
    L.R = put(16*size_t(A)*N*W);
    L.S = put(16*size_t(A)*A*W);          // largest spectrum: (A, A, W) c128
    L.idx = put(sizeof(int32_t)*size_t(A));
    L.infid = put(8*size_t(A)*A);
    L.end = o;
    return L;
}

}  // namespace

struct ffk_resident {
    double t_stage = 0, t_enqueue = 0, t_wait = 0;   // seconds, last pass (host clock)
    int device = -1;
    int G = 0, d = 0, W = 0, N = 0, A = 0;
    bool valid = false;
    Block dev = {nullptr, 0, -1}, pin = {nullptr, 0, -1};
    ResidentLayout L = {};
};

extern "C" {

int ffk_resident_create(ffk_resident** out) {
    FFK_REQUIRE(out, "NULL argument");
    *out = new (std::nothrow) ffk_resident();
    FFK_REQUIRE(*out, "out of host memory");
    return FFK_OK;
}

int ffk_resident_destroy(ffk_resident* r) {
    if (!r) return FFK_OK;
    g_dev_pool.give(r->dev);
    g_pin_pool.give(r->pin);
    delete r;
    return FFK_OK;
}

int ffk_resident_release_pools(void) {
    if (int rc = g_dev_pool.release()) return rc;
    return g_pin_pool.release();
}

}  // extern "C"

namespace {

// H[g] = sum_i c_coeffs[i, g] c_opers[i]  (pulse_sequence.py:1300-1302, 'ijk,il->ljk'), summed in
// operator order
__global__ void assemble_hamiltonian_kernel(const cplx* __restrict__ opers, const double* __restrict__ coeffs,
                                            int n_c, int G, int dd, cplx* __restrict__ H) {
    const size_t e = static_cast<size_t>(blockIdx.x)*blockDim.x + threadIdx.x;
    if (e >= static_cast<size_t>(G)*dd) return;
    const int g = static_cast<int>(e / dd), k = static_cast<int>(e % dd);
    cplx acc = {0.0, 0.0};
    for (int i = 0; i < n_c; ++i) {
        const double c = coeffs[static_cast<size_t>(i)*G + g];
        const cplx o = opers[i*dd + k];
        acc.re = fma(c, o.re, acc.re);
        acc.im = fma(c, o.im, acc.im);
    }
    H[e] = acc;
}

// Captured resident passes.  The user-facing call builds a new PulseSequence (and handle) per pulse,
// but the block pools hand the same device / pinned blocks out again and the arena does not move:
// for a given shape the enqueue is then the SAME sequence of copies and launches on the same
// addresses, call after call.  It is captured once as a hipGraph and replayed with one launch
// (H2D of the packed inputs, up to 7 kernels, D2H of the outputs: 0.037 -> 0.012 ms of host time per
// call at config 2).  Key = everything the enqueue depends on; an entry whose addresses are no
// longer handed out simply never matches again and is evicted in turn (8 entries).
struct ResidentGraphKey {
    int dev, G, d, W, N, A, n_c, on_device;
    int s_ndim, n_idx, d_inf;                        // the integral riding in the pass (0: none)
    int fault_slot;                                  // the capturing thread's fault word is baked into the kernel arguments
    const void *dp, *hp, *ws;
    hipStream_t stream;
    unsigned long long epoch;
    bool operator==(const ResidentGraphKey& o) const {
        return dev == o.dev && G == o.G && d == o.d && W == o.W && N == o.N && A == o.A && n_c == o.n_c &&
               s_ndim == o.s_ndim && n_idx == o.n_idx && d_inf == o.d_inf && fault_slot == o.fault_slot &&
               on_device == o.on_device && dp == o.dp && hp == o.hp && ws == o.ws && stream == o.stream &&
               epoch == o.epoch;
    }
};
struct ResidentGraph {
    ResidentGraphKey key;
    hipGraphExec_t exec = nullptr;
    ffk_stats stats;
    unsigned long long used = 0;
};
constexpr int kResidentGraphs = 8;
ResidentGraph g_resident_graphs[kResidentGraphs];     // guarded by g_arena.mu (held by resident_pass)
unsigned long long g_resident_graph_clock = 0;
// A pass is captured on the SECOND sighting of its key only: the key contains the pool blocks'
// addresses, and a caller that keeps its pulses alive (a gate set, a list of pulses) never gets the
// same blocks back -- every call would pay capture + instantiate + destroy and evict the graphs
// that do repeat.  The first sighting is enqueued call by call and remembered here.
constexpr int kResidentSeen = 32;
ResidentGraphKey g_resident_seen[kResidentSeen];
bool g_resident_seen_valid[kResidentSeen] = {};
int g_resident_seen_next = 0;
bool resident_key_seen_before(const ResidentGraphKey& key) {
    for (int i = 0; i < kResidentSeen; ++i)
        if (g_resident_seen_valid[i] && g_resident_seen[i] == key) return true;
    g_resident_seen[g_resident_seen_next] = key;
    g_resident_seen_valid[g_resident_seen_next] = true;
    g_resident_seen_next = (g_resident_seen_next + 1) % kResidentSeen;
    return false;
}
bool resident_graphs_enabled() {
    static const bool on = [] {
        const char* e = std::getenv("FFK_RESIDENT_GRAPH");
        return e == nullptr || e[0] != '0';
    }();
    return on;
}

// One resident pass; the Hamiltonian either given (G, d, d) or as control operators and
// amplitudes, in which case only the amplitudes cross PCIe (8 n_c B per segment instead of
// 16 d^2) and the sum runs on the device.
// Optionally the infidelity integral rides in the same pass (spectrum != NULL): ff.infidelity on a
// pulse with nothing cached is then ONE round trip to the device instead of two.
int resident_pass(ffk_resident* r, const double* hamiltonian, const double* c_opers, int n_c,
                  const double* c_coeffs, const double* dt, const double* t, int G, int d,
                  const double* omega, int W, const double* basis, int N, const double* n_opers, int A,
                  const double* n_coeffs, double** eigvals, double** eigvecs, double** propagators,
                  double** filter_function, const double* spectrum = nullptr, int s_ndim = 0,
                  int spectrum_is_real = 0, const int32_t* idx = nullptr, int n_idx = 0, int d_inf = 0,
                  double* infid = nullptr) {
    FFK_REQUIRE(r, "NULL handle");
    FFK_REQUIRE(d_templated_ok(d), "unsupported dimension d=%d (need 2 <= d <= %d)", d, FFK_MAX_D_TEMPLATED);
    FFK_REQUIRE(W >= 1 && N >= 1 && A >= 1 && G >= 1, "empty axis: W=%d N=%d A=%d G=%d", W, N, A, G);
    FFK_REQUIRE(hamiltonian || (c_opers && c_coeffs && n_c >= 1), "NULL argument");
    FFK_REQUIRE(dt && t && omega && basis && n_opers && n_coeffs, "NULL argument");
    FFK_REQUIRE(eigvals && eigvecs && propagators && filter_function, "NULL output argument");
    r->valid = false;
    int dev = 0;
    FFK_HIP(hipGetDevice(&dev));
    const ResidentLayout L = resident_layout(G, d, W, N, A);
    // spectrum (as c128), idx and the integrals live behind the outputs in the pinned block: the kernel
    // reads and writes them there (mapped memory), nothing extra crosses PCIe by copy
    size_t o_spec = 0, o_idx = 0, o_out = 0, pin_need = L.outputs_end, n_out = 0, s_rows = 0;
    if (spectrum) {
        FFK_REQUIRE(idx && infid && s_ndim >= 1 && s_ndim <= 3 && n_idx >= 1 && n_idx <= A && d_inf >= 1 && W >= 2,
                    "bad spectrum arguments");
        s_rows = s_ndim == 1 ? 1 : (s_ndim == 2 ? size_t(n_idx) : size_t(n_idx)*n_idx);
        n_out = s_ndim == 3 ? size_t(n_idx)*n_idx : size_t(n_idx);
        o_spec = align_up(L.outputs_end);
        o_idx = o_spec + align_up(16*s_rows*W);
        o_out = o_idx + align_up(sizeof(int32_t)*size_t(n_idx));
        pin_need = o_out + align_up(8*n_out);
    }
    if (r->device != dev || r->dev.size < L.end || r->pin.size < pin_need) {
        g_dev_pool.give(r->dev);
        g_pin_pool.give(r->pin);
        r->dev = r->pin = Block{nullptr, 0, -1};
        if (int rc = g_dev_pool.take(L.end, dev, &r->dev)) return rc;
        if (int rc = g_pin_pool.take(pin_need, dev, &r->pin)) return rc;
        r->device = dev;
    }
    r->G = G; r->d = d; r->W = W; r->N = N; r->A = A; r->L = L;
    unsigned char* hp = static_cast<unsigned char*>(r->pin.ptr);
    unsigned char* dp = static_cast<unsigned char*>(r->dev.ptr);
    const size_t dd = size_t(d)*d;
    const auto clock0 = std::chrono::steady_clock::now();
    // controls travel in the slot of the Hamiltonian they replace (if they fit: always, but for
    // one- or two-segment pulses with many control operators, which are summed here instead)
    const size_t ctrl_opers = 16*size_t(hamiltonian ? 0 : n_c)*dd;
    const size_t ctrl_bytes = ctrl_opers + 8*size_t(hamiltonian ? 0 : n_c)*G;
    const bool on_device = !hamiltonian && ctrl_bytes <= 16*size_t(G)*dd;
    if (hamiltonian) {
        std::memcpy(hp + L.H, hamiltonian, 16*size_t(G)*dd);
    } else if (on_device) {
        std::memcpy(hp + L.H, c_opers, ctrl_opers);
        std::memcpy(hp + L.H + ctrl_opers, c_coeffs, 8*size_t(n_c)*G);
    } else {
        double* H = reinterpret_cast<double*>(hp + L.H);
        for (int g = 0; g < G; ++g)
            for (size_t k = 0; k < dd; ++k) {
                double re = 0.0, im = 0.0;
                for (int i = 0; i < n_c; ++i) {
                    const double c = c_coeffs[size_t(i)*G + g];
                    re = std::fma(c, c_opers[2*(i*dd + k)], re);
                    im = std::fma(c, c_opers[2*(i*dd + k) + 1], im);
                }
                H[2*(g*dd + k)] = re;
                H[2*(g*dd + k) + 1] = im;
            }
    }
    std::memcpy(hp + L.dt, dt, 8*size_t(G));
    std::memcpy(hp + L.t, t, 8*size_t(G + 1));
    std::memcpy(hp + L.omega, omega, 8*size_t(W));
    std::memcpy(hp + L.basis, basis, 16*size_t(N)*dd);
    std::memcpy(hp + L.n_opers, n_opers, 16*size_t(A)*dd);
    std::memcpy(hp + L.n_coeffs, n_coeffs, 8*size_t(A)*G);
    if (spectrum) {
        double* hs = reinterpret_cast<double*>(hp + o_spec);
        if (spectrum_is_real) {
            for (size_t i = 0; i < s_rows*W; ++i) { hs[2*i] = spectrum[i]; hs[2*i + 1] = 0.0; }
        } else {
            std::memcpy(hs, spectrum, 16*s_rows*W);
        }
        std::memcpy(hp + o_idx, idx, sizeof(int32_t)*size_t(n_idx));
    }
    hipStream_t s;
    if (int rc = resident_stream(&s)) return rc;
    // scratch of the pass from the shared arena (held only for the duration of this call)
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t wsb = ffk_pipeline_workspace_bytes(W, N, A, G, d, 0, 0);
    const size_t hsb = on_device ? align_up(16*size_t(G)*dd) : 0;
    const size_t iwsb = spectrum ? align_up(ffk_infidelity_workspace_bytes(W, n_idx, s_ndim)) : 0;
    void* ws;
    if (int rc = arena_reserve(wsb + hsb + iwsb, &ws)) return rc;
    StreamDrain drain{s};      // (the successful path has synchronised already: a no-op then)
    const auto clock1 = std::chrono::steady_clock::now();
    auto dptr = [dp](size_t off) { return reinterpret_cast<double*>(dp + off); };
    const bool count_on_host = ffk::eigh_fail_count_supported(d) && ffk::use_fused_front(G, d);
    // copies in, kernels, copies out: on `s`, no synchronisation
    auto enqueue = [&]() -> int {
        const double* Hdev = dptr(L.H);
        const bool sum_in_eigensolver = on_device && count_on_host;     // (compiled d, fused front)
        if (on_device && !sum_in_eigensolver) {
            // the controls first, so that the sum runs while the rest of the inputs is still in flight
            FFK_HIP(hipMemcpyAsync(dp + L.H, hp + L.H, ctrl_bytes, hipMemcpyHostToDevice, s));
            cplx* Hsum = reinterpret_cast<cplx*>(static_cast<unsigned char*>(ws) + wsb);
            const size_t n = size_t(G)*dd;
            hipLaunchKernelGGL(assemble_hamiltonian_kernel, dim3(static_cast<unsigned>((n + 255)/256)), dim3(256),
                               0, s, reinterpret_cast<const cplx*>(dp + L.H),
                               reinterpret_cast<const double*>(dp + L.H + ctrl_opers), n_c, G, d*d, Hsum);
            FFK_HIP(hipGetLastError());
            FFK_HIP(hipMemcpyAsync(dp + L.dt, hp + L.dt, L.inputs_end - L.dt, hipMemcpyHostToDevice, s));
            Hdev = reinterpret_cast<const double*>(Hsum);
        } else {
            // one copy; with the controls in the Hamiltonian's slot the eigensolver sums them itself
            FFK_HIP(hipMemcpyAsync(dp, hp, L.inputs_end, hipMemcpyHostToDevice, s));
        }
        PassOptions opt;
        if (sum_in_eigensolver)
            opt.eigh_controls = {reinterpret_cast<const cplx*>(dp + L.H),
                                 reinterpret_cast<const double*>(dp + L.H + ctrl_opers), n_c};
        // the eigensolver counts its flagged segments straight into the pinned block's status word (zeroed below,
        // before the launch): no memset, no counting kernel, and the word stays out of the copy back
        opt.eigh_fail_count = count_on_host ? reinterpret_cast<int*>(hp + L.status) : nullptr;
        if (int rc = pipeline_dev_impl(Hdev, dptr(L.dt), dptr(L.t), G, d, dptr(L.omega), W, dptr(L.basis), N,
                                       dptr(L.n_opers), A, dptr(L.n_coeffs), nullptr, 0, nullptr, 0, dptr(L.D),
                                       dptr(L.V), dptr(L.Q), dptr(L.R), dptr(L.F), nullptr, ws, wsb, s, opt))
            return rc;
        if (!count_on_host)
            if (int rc = ffk_eigensolver_status_dev(ws, wsb, G, d, reinterpret_cast<int32_t*>(dp + L.status), s))
                return rc;
        if (spectrum) {
            opt.infid_spectrum_on_host = true;     // spectrum and idx are read from the pinned block
            if (int rc = infidelity_dev_impl(dptr(L.F), A, W, reinterpret_cast<const double*>(hp + o_spec), s_ndim,
                                             dptr(L.omega), reinterpret_cast<const int32_t*>(hp + o_idx), n_idx,
                                             d_inf, reinterpret_cast<double*>(hp + o_out),
                                             static_cast<unsigned char*>(ws) + wsb + hsb, iwsb, s, opt))
                return rc;
        }
        FFK_HIP(hipMemcpyAsync(hp + L.D, dp + L.D, (count_on_host ? L.status : L.outputs_end) - L.D,
                               hipMemcpyDeviceToHost, s));
        return FFK_OK;
    };
    if (count_on_host) *reinterpret_cast<volatile int32_t*>(hp + L.status) = 0;
    const ResidentGraphKey key{dev, G, d, W, N, A, hamiltonian ? 0 : n_c, on_device ? 1 : 0,
                               spectrum ? s_ndim : 0, spectrum ? n_idx : 0, spectrum ? d_inf : 0,
                               kernel_fault_slot_for_selftest(), dp, hp, ws, s, g_knob_epoch.load()};
    ResidentGraph* hit = nullptr;
    ResidentGraph* victim = &g_resident_graphs[0];
    if (resident_graphs_enabled()) {
        for (ResidentGraph& e : g_resident_graphs) {
            if (e.exec && e.key == key) hit = &e;
            if (e.used < victim->used) victim = &e;
        }
    }
    bool enqueued = false;
    if (hit) {
        if (hipGraphLaunch(hit->exec, s) == hipSuccess) {
            hit->used = ++g_resident_graph_clock;
            g_stats = hit->stats;
            enqueued = true;
        } else {
            (void)hipGetLastError();
            (void)hipGraphExecDestroy(hit->exec);
            hit->exec = nullptr;
            hit->used = 0;
        }
    } else if (resident_graphs_enabled() && resident_key_seen_before(key) &&
               hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) == hipSuccess) {
        // second pass of this shape on these blocks: capture it, then launch the capture
        const int rc = enqueue();
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(s, &graph);
        if (rc != FFK_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            return rc;
        }
        hipGraphExec_t exec = nullptr;
        if (ce == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess &&
            hipGraphLaunch(exec, s) == hipSuccess) {
            if (victim->exec) (void)hipGraphExecDestroy(victim->exec);
            victim->key = key;
            victim->exec = exec;
            victim->stats = g_stats;
            victim->used = ++g_resident_graph_clock;
            enqueued = true;
        } else {
            if (exec) (void)hipGraphExecDestroy(exec);
            (void)hipGetLastError();
        }
        if (graph) (void)hipGraphDestroy(graph);
    } else {
        (void)hipGetLastError();
    }
    if (!enqueued)
        if (int rc = enqueue()) return rc;
    const auto clock2 = std::chrono::steady_clock::now();
    FFK_HIP(hipStreamSynchronize(s));
    const auto clock3 = std::chrono::steady_clock::now();
    r->t_stage = std::chrono::duration<double>(clock1 - clock0).count();
    r->t_enqueue = std::chrono::duration<double>(clock2 - clock1).count();
    r->t_wait = std::chrono::duration<double>(clock3 - clock2).count();
    const int32_t failed = *reinterpret_cast<const int32_t*>(hp + L.status);
    if (failed != 0)
        return fail(FFK_ENOCONV, "Jacobi eigensolver did not converge for %d segment(s)", int(failed));
    if (int rc = kernel_fault_status()) return rc;
    *eigvals = reinterpret_cast<double*>(hp + L.D);
    *eigvecs = reinterpret_cast<double*>(hp + L.V);
    *propagators = reinterpret_cast<double*>(hp + L.Q);
    *filter_function = reinterpret_cast<double*>(hp + L.F);
    if (spectrum) std::memcpy(infid, hp + o_out, 8*n_out);
    r->valid = true;
    return FFK_OK;
}

}  // namespace

extern "C" {

int ffk_resident_filter_function_infidelity(ffk_resident* r, const double* c_opers, int n_cops,
                                            const double* c_coeffs, const double* dt, const double* t, int G,
                                            int d, const double* omega, int W, const double* basis, int N,
                                            const double* n_opers, int A, const double* n_coeffs,
                                            const double* spectrum, int s_ndim, int spectrum_is_real,
                                            const int32_t* idx, int n_idx, int d_infidelity, double** eigvals,
                                            double** eigvecs, double** propagators, double** filter_function,
                                            double* infidelity) {
    FFK_REQUIRE(spectrum && idx && infidelity, "NULL argument");
    return resident_pass(r, nullptr, c_opers, n_cops, c_coeffs, dt, t, G, d, omega, W, basis, N, n_opers, A,
                         n_coeffs, eigvals, eigvecs, propagators, filter_function, spectrum, s_ndim,
                         spectrum_is_real, idx, n_idx, d_infidelity, infidelity);
}

int ffk_resident_filter_function(ffk_resident* r, const double* hamiltonian, const double* dt,
                                 const double* t, int G, int d, const double* omega, int W,
                                 const double* basis, int N, const double* n_opers, int A,
                                 const double* n_coeffs, double** eigvals, double** eigvecs,
                                 double** propagators, double** filter_function) {
    FFK_REQUIRE(hamiltonian, "NULL argument");
    return resident_pass(r, hamiltonian, nullptr, 0, nullptr, dt, t, G, d, omega, W, basis, N, n_opers, A,
                         n_coeffs, eigvals, eigvecs, propagators, filter_function);
}

int ffk_resident_filter_function_from_controls(ffk_resident* r, const double* c_opers, int n_cops,
                                               const double* c_coeffs, const double* dt,
                                               const double* t, int G, int d, const double* omega,
                                               int W, const double* basis, int N,
                                               const double* n_opers, int A, const double* n_coeffs,
                                               double** eigvals, double** eigvecs,
                                               double** propagators, double** filter_function) {
    FFK_REQUIRE(c_opers && c_coeffs && n_cops >= 1, "NULL or empty control Hamiltonian");
    return resident_pass(r, nullptr, c_opers, n_cops, c_coeffs, dt, t, G, d, omega, W, basis, N, n_opers, A,
                         n_coeffs, eigvals, eigvecs, propagators, filter_function);
}

int ffk_resident_timing(ffk_resident* r, double* seconds) {
    FFK_REQUIRE(r && seconds, "NULL argument");
    seconds[0] = r->t_stage;
    seconds[1] = r->t_enqueue;
    seconds[2] = r->t_wait;
    return FFK_OK;
}

namespace {
int on_owning_device(const ffk_resident* r) {
    int dev = -1;
    FFK_HIP(hipGetDevice(&dev));
    FFK_REQUIRE(dev == r->device, "resident result lives on device %d, current device is %d",
                r->device, dev);
    return FFK_OK;
}
}  // namespace

}  // extern "C"


extern "C" {

// ffk_concatenate_sequence for distinct pulses whose control matrices are still resident (every one
// evaluated by ffk_resident_filter_function* on the same frequency grid): the table is assembled
// by device-to-device copies, the total propagators come from the handles' host blocks, the total
// phase factors exp(i omega tau_k) are formed on the device -- per call only index, basis and tau
// cross PCIe.  With `result` (which = 0, filter_function wanted) the summed control matrix, its filter
// function and the grid STAY in that handle (control_matrix may then be NULL): the new pulse is as
// resident as its parts -- ffk_resident_control_matrix / _infidelity serve it, and it can be an
// input of the next concatenation.
int ffk_concatenate_sequence_resident(ffk_resident* const* pulses, const double* tau,
                                      const int32_t* index, const double* basis, int hermitian_basis,
                                      int T, int G, int which, double* control_matrix,
                                      double* total_propagator, double* propagators_liouville,
                                      double* filter_function, ffk_resident* result) {
    FFK_REQUIRE(pulses && tau && index && basis && total_propagator, "NULL argument");
    FFK_REQUIRE(control_matrix || result, "NULL argument");
    FFK_REQUIRE(!result || (which == 0 && filter_function), "a resident result holds the summed control "
                "matrix and its filter function");
    for (int k = 0; result && k < T; ++k) FFK_REQUIRE(pulses[k] != result, "result must not be an input");
    FFK_REQUIRE(T >= 1 && T <= 65535 && G >= 1, "empty or oversized axis: T=%d G=%d", T, G);
    FFK_REQUIRE(which == 0 || which == 1, "invalid which=%d", which);
    FFK_REQUIRE(!filter_function || which == 0, "the filter function needs the summed control matrix");
    for (int k = 0; k < T; ++k) FFK_REQUIRE(pulses[k] && pulses[k]->valid, "pulse %d has no resident result", k);
    const ffk_resident* first = pulses[0];
    const int d = first->d, A = first->A, N = first->N, W = first->W;
    for (int k = 0; k < T; ++k) {
        const ffk_resident* r = pulses[k];
        FFK_REQUIRE(r->d == d && r->A == A && r->N == N && r->W == W && r->device == first->device,
                    "pulse %d: shape (d=%d, A=%d, N=%d, W=%d) or device differs from pulse 0", k, r->d,
                    r->A, r->N, r->W);
    }
    for (int g = 0; g < G; ++g)
        FFK_REQUIRE(index[g] >= 0 && index[g] < T, "index[%d] = %d outside [0, %d)", g, index[g], T);
    if (int rc = on_owning_device(first)) return rc;
    hipStream_t s;
    if (int rc = resident_stream(&s)) return rc;
    std::lock_guard<std::mutex> lock(g_arena.mu);
    const size_t dd = size_t(d)*d;
    const size_t nU = 16*size_t(T)*dd, nP = 16*size_t(T)*W, nR1 = 16*size_t(A)*N*W, nI = 4*size_t(G);
    const size_t nB = 16*size_t(N)*dd, nT = 8*size_t(T), nX = 8*size_t(T);
    // host staging (propagators | tau | index | basis | pointers to the resident control matrices) in
    // one pinned block, one H2D.  The control matrices are read where they lie (round 3: assembling
    // a contiguous table cost T device-to-device copies per call, 24 of ~0.5 MB at config 3)
    const size_t oU = 0, oT = oU + align_up(nU), oI = oT + align_up(nT), oB = oI + align_up(nI);
    const size_t oX = oB + align_up(nB);
    const size_t stage = oX + align_up(nX);
    Block pin = {nullptr, 0, -1};
    if (int rc = g_pin_pool.take(stage, first->device, &pin)) return rc;
    unsigned char* hp = static_cast<unsigned char*>(pin.ptr);
    for (int k = 0; k < T; ++k) {
        const ffk_resident* r = pulses[k];
        const unsigned char* q = static_cast<const unsigned char*>(r->pin.ptr) + r->L.Q + 16*size_t(r->G)*dd;
        std::memcpy(hp + oU + 16*size_t(k)*dd, q, 16*dd);         // Q[-1]: the pulse's total propagator
    }
    std::memcpy(hp + oT, tau, nT);
    std::memcpy(hp + oI, index, nI);
    std::memcpy(hp + oB, basis, nB);
    for (int k = 0; k < T; ++k) {
        const unsigned char* rk = static_cast<const unsigned char*>(pulses[k]->dev.ptr) + pulses[k]->L.R;
        std::memcpy(hp + oX + 8*size_t(k), &rk, 8);
    }
    (void)nR1;
    void* base;
    int rc = arena_reserve(stage + align_up(nP) +
                           sequence_scratch_bytes(G, d, A, N, W, which, hermitian_basis != 0,
                                                  filter_function != nullptr), &base);
    if (rc) { g_pin_pool.give(pin); return rc; }
    Bump a(base, g_arena.size);
    unsigned char* dS = a.take<unsigned char>(stage);
    double* dP = a.take<double>(nP/8);
    // a result handle takes the layout of a one-segment pass: R, F and the grid in its device block,
    // (identity, total propagator) where the propagators of a pass sit in its host block
    ResidentLayout RL = {};
    double* keep_R = nullptr;
    double* keep_F = nullptr;
    if (result) {
        result->valid = false;
        RL = resident_layout(1, d, W, N, A);
        if (result->device != first->device || result->dev.size < RL.end || result->pin.size < RL.outputs_end) {
            g_dev_pool.give(result->dev);
            g_pin_pool.give(result->pin);
            result->dev = result->pin = Block{nullptr, 0, -1};
            rc = g_dev_pool.take(RL.end, first->device, &result->dev);
            if (!rc) rc = g_pin_pool.take(RL.outputs_end, first->device, &result->pin);
            if (rc) { g_pin_pool.give(pin); return rc; }
            result->device = first->device;
        }
        result->G = 1; result->d = d; result->W = W; result->N = N; result->A = A; result->L = RL;
        unsigned char* rp = static_cast<unsigned char*>(result->dev.ptr);
        keep_R = reinterpret_cast<double*>(rp + RL.R);
        keep_F = reinterpret_cast<double*>(rp + RL.F);
    }
    auto run = [&]() -> int {
        StreamDrain drain{s};
        FFK_HIP(hipMemcpyAsync(dS, hp, stage, hipMemcpyHostToDevice, s));
        const double* dOmega = reinterpret_cast<const double*>(
            static_cast<const unsigned char*>(first->dev.ptr) + first->L.omega);
        double* omega_copy = result ? reinterpret_cast<double*>(static_cast<unsigned char*>(result->dev.ptr) + RL.omega)
                                    : nullptr;
        if (int rc2 = sequence_on_device(reinterpret_cast<const double*>(dS + oU), dP, nullptr,
                                         reinterpret_cast<const int32_t*>(dS + oI),
                                         reinterpret_cast<const double*>(dS + oB), hermitian_basis, T, G, d,
                                         A, N, W, which, a, control_matrix, total_propagator,
                                         propagators_liouville, filter_function, s, keep_R, keep_F,
                                         reinterpret_cast<const cplx* const*>(dS + oX),
                                         reinterpret_cast<const double*>(dS + oT), dOmega, omega_copy))
            return rc2;
        FFK_HIP(hipStreamSynchronize(s));
        return FFK_OK;
    };
    rc = run();
    g_pin_pool.give(pin);
    if (!rc && result) {
        double* q = reinterpret_cast<double*>(static_cast<unsigned char*>(result->pin.ptr) + RL.Q);
        for (size_t e = 0; e < dd; ++e) {
            q[2*e] = (e / d == e % d) ? 1.0 : 0.0;
            q[2*e + 1] = 0.0;
        }
        std::memcpy(q + 2*dd, total_propagator, 16*dd);
        result->t_stage = result->t_enqueue = result->t_wait = 0.0;
        result->valid = true;
    }
    return rc;
}

int ffk_resident_control_matrix(ffk_resident* r, double* control_matrix) {
    FFK_REQUIRE(r && r->valid, "no resident result");
    FFK_REQUIRE(control_matrix, "NULL argument");
    if (int rc = on_owning_device(r)) return rc;
    hipStream_t s;
    if (int rc = resident_stream(&s)) return rc;
    const unsigned char* dp = static_cast<const unsigned char*>(r->dev.ptr);
    FFK_HIP(hipMemcpyAsync(control_matrix, dp + r->L.R, 16*size_t(r->A)*r->N*r->W,
                           hipMemcpyDeviceToHost, s));
    FFK_HIP(hipStreamSynchronize(s));
    return FFK_OK;
}

int ffk_resident_control_matrix_dev(ffk_resident* r, const double** control_matrix,
                                    const double** filter_function, const double** omega) {
    FFK_REQUIRE(r && r->valid, "no resident result");
    const unsigned char* dp = static_cast<const unsigned char*>(r->dev.ptr);
    if (control_matrix) *control_matrix = reinterpret_cast<const double*>(dp + r->L.R);
    if (filter_function) *filter_function = reinterpret_cast<const double*>(dp + r->L.F);
    if (omega) *omega = reinterpret_cast<const double*>(dp + r->L.omega);
    return FFK_OK;
}

int ffk_resident_infidelity(ffk_resident* r, const double* spectrum, int s_ndim, int spectrum_is_real,
                            const int32_t* idx, int n_idx, int d, double* infid) {
    FFK_REQUIRE(r && r->valid, "no resident result");
    FFK_REQUIRE(spectrum && idx && infid, "NULL argument");
    FFK_REQUIRE(s_ndim >= 1 && s_ndim <= 3 && n_idx >= 1 && n_idx <= r->A && d >= 1, "bad spectrum arguments");
    if (int rc = on_owning_device(r)) return rc;
    const int W = r->W, A = r->A;
    const size_t rows = s_ndim == 1 ? 1 : (s_ndim == 2 ? size_t(n_idx) : size_t(n_idx)*n_idx);
    const size_t n_out = s_ndim == 3 ? size_t(n_idx)*n_idx : size_t(n_idx);
    unsigned char* hp = static_cast<unsigned char*>(r->pin.ptr);
    unsigned char* dp = static_cast<unsigned char*>(r->dev.ptr);
    const ResidentLayout& L = r->L;
    // stage spectrum (as c128), idx and the result in the pinned input region (free after the pass).
    // Round 3: the kernel reads spectrum and idx FROM that pinned block and writes the integrals INTO
    // it (pinned host memory is mapped into the device's address space): one launch and one
    // synchronisation instead of two H2D copies, the launch, a D2H copy into pageable memory and
    // the synchronisation -- the spectrum is read once (64 KB over PCIe at config 2)
    const size_t s_bytes = 16*rows*W;
    const size_t o_idx = align_up(s_bytes), o_out = o_idx + align_up(sizeof(int32_t)*size_t(n_idx));
    const size_t stage = o_out + align_up(8*n_out);
    hipStream_t s;
    if (int rc = resident_stream(&s)) return rc;
    const bool fits = stage <= L.inputs_end;
    Block extra = {nullptr, 0, -1};
    unsigned char* stage_ptr = hp;
    if (!fits) {
        if (int rc = g_pin_pool.take(stage, r->device, &extra)) return rc;
        stage_ptr = static_cast<unsigned char*>(extra.ptr);
    }
    double* hs = reinterpret_cast<double*>(stage_ptr);
    if (spectrum_is_real) {
        for (size_t i = 0; i < rows*W; ++i) { hs[2*i] = spectrum[i]; hs[2*i + 1] = 0.0; }
    } else {
        std::memcpy(hs, spectrum, s_bytes);
    }
    std::memcpy(stage_ptr + o_idx, idx, sizeof(int32_t)*size_t(n_idx));
    int rc = FFK_OK;
    {
        std::lock_guard<std::mutex> lock(g_arena.mu);
        const size_t iwsb = ffk_infidelity_workspace_bytes(W, n_idx, s_ndim);
        void* iws;
        rc = arena_reserve(iwsb, &iws);
        if (!rc) {
            PassOptions opt;
            opt.infid_spectrum_on_host = true;
            rc = infidelity_dev_impl(reinterpret_cast<const double*>(dp + L.F), A, W,
                                     reinterpret_cast<const double*>(stage_ptr), s_ndim,
                                     reinterpret_cast<const double*>(dp + L.omega),
                                     reinterpret_cast<const int32_t*>(stage_ptr + o_idx), n_idx, d,
                                     reinterpret_cast<double*>(stage_ptr + o_out), iws, iwsb, s, opt);
        }
        if (!rc) {
            const hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) rc = fail(FFK_EHIP, "infidelity failed: %s", hipGetErrorString(e));
            else std::memcpy(infid, stage_ptr + o_out, 8*n_out);
        }
    }
    g_pin_pool.give(extra);
    return rc;
}

}  // extern "C"

#if defined(FFK_HOST_SANITIZE)
// ---------------------------------------------------------------------------------------------
// Host-logic self test for the sanitizer variant: arena growth, block-pool reuse and eviction, every
// workspace layout sliced with the size its *_workspace_bytes query reports and each slice written
// end to end (an overrun of a slice or of the reservation is a heap-buffer-overflow under ASan).
// ---------------------------------------------------------------------------------------------
#include <cstring>
#include <random>
extern "C" int ffk_selftest_host(int rounds, unsigned seed, char* report, int report_len) {
    if (rounds < 0) {
        // negative control: a deliberate one-byte overrun, which the sanitizer must report
        volatile unsigned char* p = static_cast<unsigned char*>(std::malloc(16));
        p[16] = 1;
        std::free(const_cast<unsigned char*>(p));
        return 0;
    }
    std::mt19937 rng(seed);
    auto pick = [&](int lo, int hi) { return lo + static_cast<int>(rng() % static_cast<unsigned>(hi - lo + 1)); };
    long checked = 0;
    auto touch = [&](void* p, size_t n) {
        if (p && n) {
            std::memset(p, 0xA5, n);
            ++checked;
        }
    };
    for (int r = 0; r < rounds; ++r) {
        const int d = pick(2, FFK_MAX_D_TEMPLATED), G = pick(1, 300), A = pick(1, 9), W = pick(1, 700);
        const int N = pick(1, d*d);
        // (a) arena: reserve, write all of it, grow, shrink requests
        void* base = nullptr;
        const size_t want = size_t(pick(1, 1 << 20))*pick(1, 8);
        {
            std::lock_guard<std::mutex> lock(g_arena.mu);
            if (arena_reserve(want, &base) != FFK_OK) return -1;
            touch(base, g_arena.size);
        }
        // (a') the kernels' fault words: one per host thread -- a word set for this thread is seen (and cleared) by this
        // thread's check and by no other thread's; the word the launchers would hand their kernels is this thread's
        if (r == 0) {
            int* mine = ffk::kernel_fault_word();
            if (mine == nullptr) return -20;
            if (kernel_fault_peek(false) != 0) return -21;
            *static_cast<volatile int*>(mine) = ffk::kernel_fault_code_for_selftest();
            int other_saw = -1, other_slot = -1;
            int* others = nullptr;
            std::thread peer([&] {
                others = ffk::kernel_fault_word();
                other_slot = kernel_fault_slot_for_selftest();
                other_saw = kernel_fault_peek(true);          // must not see (or clear) this thread's fault
            });
            peer.join();
            if (others == nullptr || others == mine || other_slot == kernel_fault_slot_for_selftest()) return -22;
            if (other_saw != 0) return -23;
            if (kernel_fault_stale() != FFK_EKERNEL) return -24;      // reported to the thread that owns it, once
            if (kernel_fault_peek(false) != 0 || kernel_fault_status() != FFK_OK) return -25;
            if (ffk::kernel_fault_word() != mine) return -26;          // a thread keeps its word
        }
        // (b) control-matrix workspace: the slices of ffk_control_matrix_dev
        {
            const ffk::AccumGeometry geo = ffk::accumulate_geometry(W, A, G, d, 0);
            const size_t bytes = ffk_control_matrix_workspace_bytes(W, N, A, G, d);
            if (bytes < ctrl_ws_bytes(W, N, A, G, d, geo.chunks)) return -2;
            void* ws = std::malloc(bytes);
            Bump b(ws, bytes);
            double* segtab = b.take<double>(size_t(G)*ffk::seg_stride(d));
            cplx* Tc = b.take<cplx>(size_t(G)*d*d);
            cplx* ops = b.take<cplx>(size_t(G)*(1 + A)*d*d);
            cplx* Ypart = b.take<cplx>(size_t(geo.chunks)*A*d*d*W);
            cplx* Bt = b.take<cplx>(size_t(A)*d*d*W);
            void* ews = b.take<unsigned char>(ffk::expand_workspace_bytes(N, d));
            if (!segtab || !Tc || !ops || !Ypart || !Bt || !ews) { std::free(ws); return -3; }
            touch(segtab, sizeof(double)*size_t(G)*ffk::seg_stride(d));
            touch(Tc, sizeof(cplx)*size_t(G)*d*d);
            touch(ops, sizeof(cplx)*size_t(G)*(1 + A)*d*d);
            touch(Ypart, sizeof(cplx)*size_t(geo.chunks)*A*d*d*W);
            touch(Bt, sizeof(cplx)*size_t(A)*d*d*W);
            touch(ews, ffk::expand_workspace_bytes(N, d));
            int *nnz, *rows;
            cplx* vals;
            ffk::expand_workspace_slices(ews, N, d, &nnz, &rows, &vals);
            touch(nnz, sizeof(int)*N);
            touch(rows, sizeof(int)*size_t(N)*d*d);
            touch(vals, sizeof(cplx)*size_t(N)*d*d);
            std::free(ws);
        }
        // (c) diagonalize + pipeline workspaces
        {
            const size_t dwsb = ffk_diagonalize_workspace_bytes(G, d);
            void* ws = std::malloc(dwsb);
            const DiagWs w = slice_diag_ws(ws, dwsb, G, d);
            touch(w.status, sizeof(int)*G);
            touch(w.seg_prop, sizeof(cplx)*size_t(G)*d*d);
            touch(w.qloc, sizeof(cplx)*size_t(G + 1)*d*d);
            std::free(ws);
            const int n_idx = pick(1, A), s_ndim = pick(1, 3);
            const size_t pb = ffk_pipeline_workspace_bytes(W, N, A, G, d, n_idx, s_ndim);
            if (pb < dwsb + ffk_control_matrix_workspace_bytes(W, N, A, G, d)) return -4;
            void* pws = std::malloc(pb);
            Bump b(pws, pb);
            void* a1 = b.take<unsigned char>(dwsb);
            void* a2 = b.take<unsigned char>(ffk_control_matrix_workspace_bytes(W, N, A, G, d));
            double* D = b.take<double>(size_t(G)*d);
            double* V = b.take<double>(2*size_t(G)*d*d);
            double* Q = b.take<double>(2*size_t(G + 1)*d*d);
            double* R = b.take<double>(2*size_t(A)*N*W);
            double* F = b.take<double>(2*size_t(A)*A*W);
            void* iws = b.take<unsigned char>(ffk_infidelity_workspace_bytes(W, n_idx, s_ndim));
            if (!a1 || !a2 || !D || !V || !Q || !R || !F || !iws) { std::free(pws); return -5; }
            touch(F, 16*size_t(A)*A*W);
            touch(iws, ffk_infidelity_workspace_bytes(W, n_idx, s_ndim));
            std::free(pws);
        }
        // (d) sequence scratch of the concatenation entry points
        {
            const int T = pick(1, 30), Gs = pick(1, 1200), which = pick(0, 1);
            const bool herm = pick(0, 1) != 0, wantF = which == 0 && pick(0, 1);
            const int d2 = pick(2, 4), N2 = d2*d2, A2 = pick(1, 3), W2 = pick(1, 300);
            (void)T;
            const size_t sb = sequence_scratch_bytes(Gs, d2, A2, N2, W2, which, herm, wantF);
            void* ws = std::malloc(sb);
            Bump a(ws, sb);
            const size_t dd = size_t(d2)*d2;
            const int nl = Gs > 1 ? Gs - 1 : 1;
            cplx* dSeq = a.take<cplx>(size_t(Gs)*dd);
            cplx* dQ = a.take<cplx>(size_t(Gs + 1)*dd);
            double* dL = a.take<double>((herm ? 1 : 2)*size_t(nl)*N2*N2);
            double* dO = a.take<double>(2*(which ? size_t(Gs) : 1)*A2*N2*W2);
            void* w1 = a.take<unsigned char>(ffk::scan_workspace_bytes(Gs, d2));
            void* w2 = a.take<unsigned char>(ffk::liouville_workspace_bytes(nl, d2, N2));
            void* w3 = a.take<unsigned char>(ffk_control_matrix_from_atomic_workspace_bytes(Gs, A2, N2, W2));
            double* dF = wantF ? a.take<double>(2*size_t(A2)*A2*W2) : nullptr;
            if (!dSeq || !dQ || !dL || !dO || !w1 || !w2 || !w3 || (wantF && !dF)) { std::free(ws); return -6; }
            touch(dSeq, 16*size_t(Gs)*dd);
            touch(dQ, 16*size_t(Gs + 1)*dd);
            touch(dL, (herm ? 8 : 16)*size_t(nl)*N2*N2);
            touch(dO, 16*(which ? size_t(Gs) : 1)*A2*N2*W2);
            touch(w3, ffk_control_matrix_from_atomic_workspace_bytes(Gs, A2, N2, W2));
            if (dF) touch(dF, 16*size_t(A2)*A2*W2);
            std::free(ws);
        }
        // (e) block pools: take / write / give in random order, past the eviction bound
        {
            std::vector<Block> held;
            for (int k = 0; k < 40; ++k) {
                if (held.empty() || pick(0, 2)) {
                    Block b = {nullptr, 0, -1};
                    BlockPool& pool = pick(0, 1) ? g_dev_pool : g_pin_pool;
                    const size_t bytes = size_t(pick(1, 1 << 18));
                    if (pool.take(bytes, 0, &b) != FFK_OK || b.size < bytes) return -7;
                    touch(b.ptr, b.size);
                    b.device = (&pool == &g_dev_pool) ? 0 : -1;
                    held.push_back(b);
                } else {
                    const int i = pick(0, int(held.size()) - 1);
                    (held[i].device == 0 ? g_dev_pool : g_pin_pool).give(held[i]);
                    held.erase(held.begin() + i);
                }
            }
            for (const Block& b : held) (b.device == 0 ? g_dev_pool : g_pin_pool).give(b);
        }
        // (f) resident layout: offsets ascending, inside the block
        {
            const ResidentLayout RL = resident_layout(G, d, W, N, A);
            if (!(RL.inputs_end <= RL.D && RL.outputs_end <= RL.R && RL.R < RL.end && RL.F + 16*size_t(A)*A*W <= RL.end))
                return -8;
            ffk_resident* h = nullptr;
            if (ffk_resident_create(&h) != FFK_OK) return -9;
            if (ffk_resident_destroy(h) != FFK_OK) return -10;
        }
    }
    if (g_dev_pool.release() != FFK_OK || g_pin_pool.release() != FFK_OK) return -11;
    if (ffk_release_arena() != FFK_OK) return -12;
    if (report && report_len > 0)
        snprintf(report, report_len, "%d rounds, %ld regions written end to end, pools and arena released", rounds, checked);
    return 0;
}
#endif
