// Torch extension binding of the SMIN hot path: TORCH_LIBRARY(smin_hip, ...) over the C ABI of include/smin_hip.h.
//
// The reference's operator boundary is the nn.Module surface of models.py; below it the drop-in runs as
// ONE library call per forward: smin_hip::smin_forward takes the six forward arguments of SMIN.forward
// (reference models.py:367) plus the module's parameters and builds the whole autograd graph in C++ --
// torch::autograd::Function nodes around the HIP entry points, one per reference module body, so the backward pass runs
// on the autograd engine's thread without the interpreter (DistributedDataParallel hooks fire as usual).
// The Python host (video-moment-localization_amd/functional.py, modules.py) binds the same C ABI with ctypes; it
// serves the stand-alone sub-module seams and remains as a second host for the in-model path (SMIN.native_host = False),
// kernel for kernel the same launches.  torch types appear only in this file; libsmin_hip.so knows pointers and sizes.
//
// Only the in-model fast path lives here: the content stream on a mask-driven cell list.
#include <ATen/ATen.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <hip/hip_runtime_api.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>
#include <ATen/core/dispatch/Dispatcher.h>
#include <ATen/core/stack.h>

#include <cmath>
#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "smin_hip.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;
using HStream = c10::hip::HIPStream;

#define SMIN_CK(call)                                                                           \
    do {                                                                                        \
        const int rc__ = (call);                                                                \
        TORCH_CHECK(rc__ == 0, "smin_hip: " #call " failed with code ", rc__,                   \
                    rc__ < -1000 ? " (argument rejected at csrc line " + std::to_string(-rc__ - 1000) + ")" : std::string()); \
    } while (0)

// ---------------------------------------------------------------- small helpers
inline const float* fp(const Tensor& t) { return t.defined() ? t.const_data_ptr<float>() : nullptr; }
inline float* fpm(const Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
inline const uint16_t* hp16(const Tensor& t) { return reinterpret_cast<const uint16_t*>(t.const_data_ptr()); }     // bf16 tensors as bit patterns
inline const int32_t* ip(const Tensor& t) { return t.const_data_ptr<int32_t>(); }
inline void* cur() { return (void*)c10::hip::getCurrentHIPStream().stream(); }
inline Tensor cont(const Tensor& t) { return t.defined() ? t.contiguous() : t; }
inline Tensor fl(const Tensor& t) { return t.scalar_type() == at::kFloat ? t : t.to(at::kFloat); }
inline int i32(int64_t v) { return static_cast<int>(v); }

struct StreamScope {                       // torch's current stream for the scope (what torch.cuda.stream(s) does)
    HStream prev;
    explicit StreamScope(HStream s) : prev(c10::hip::getCurrentHIPStream(s.device_index())) { c10::hip::setCurrentHIPStream(s); }
    ~StreamScope() { c10::hip::setCurrentHIPStream(prev); }
};

// Events for stream joins: a ring per device (an event is bound to the device it was created on), created under that device's
// guard.  A draw is consumed (recorded and waited for by hipStreamWaitEvent, or synchronised by the host) within the call that
// drew it; the ring is far longer than the draws of one forward + backward pass (~60 at three layers), so no event is
// re-recorded while a wait on its previous record is still being enqueued.
hipEvent_t next_event()
{
    constexpr size_t RING = 1024;
    static std::mutex mu;
    static std::map<int, std::pair<std::vector<hipEvent_t>, size_t>> rings;
    int dev = 0;
    TORCH_CHECK(hipGetDevice(&dev) == hipSuccess, "hipGetDevice failed");
    std::lock_guard<std::mutex> lk(mu);
    auto& ring = rings[dev];
    if (ring.first.empty()) {
        ring.first.resize(RING);
        for (auto& e : ring.first) TORCH_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess, "hipEventCreate failed");
    }
    return ring.first[ring.second++ % RING];
}
// `waiter` waits for everything queued on `on` so far (stream.wait_stream)
void wait_stream(HStream waiter, HStream on)
{
    if (waiter == on) return;
    hipEvent_t e = next_event();
    TORCH_CHECK(hipEventRecord(e, on.stream()) == hipSuccess, "hipEventRecord failed");
    TORCH_CHECK(hipStreamWaitEvent(waiter.stream(), e, 0) == hipSuccess, "hipStreamWaitEvent failed");
}
void record_stream(const Tensor& t, HStream s)
{
    if (t.defined() && t.has_storage()) c10::hip::HIPCachingAllocator::recordStream(t.storage().data_ptr(), s);
}
HStream side_stream(c10::DeviceIndex dev, int which = 0)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, HStream> streams;
    std::lock_guard<std::mutex> lk(mu);
    auto it = streams.find({(int)dev, which});
    if (it == streams.end()) {
        HStream st = c10::hip::getStreamFromPool(false, dev);
        for (auto& kv : streams)                                  // the pool hands its streams out round robin: never the same one twice
            while (kv.first.first == (int)dev && kv.second == st) st = c10::hip::getStreamFromPool(false, dev);
        it = streams.emplace(std::make_pair((int)dev, which), st).first;
    }
    return it->second;
}

// lowest-priority stream for work nobody waits for until the end of the step (weight-gradient contractions)
HStream weight_stream(c10::DeviceIndex dev)
{
    static std::mutex mu;
    static std::map<int, HStream> streams;
    std::lock_guard<std::mutex> lk(mu);
    auto it = streams.find(dev);
    if (it == streams.end()) {
        int least = 0, greatest = 0;
        TORCH_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess, "hipDeviceGetStreamPriorityRange failed");
        hipStream_t raw;
        TORCH_CHECK(hipStreamCreateWithPriority(&raw, hipStreamNonBlocking, least) == hipSuccess, "hipStreamCreateWithPriority failed");
        it = streams.emplace((int)dev, c10::hip::getStreamFromExternal(raw, dev)).first;
    }
    return it->second;
}

// persistent scratch per (device, stream): calls on one stream are stream-ordered, two streams never share scratch
struct Scratch { void* p; size_t n; Tensor hold; };
Scratch scratch(size_t nbytes, const at::Device& dev)
{
    static std::mutex mu;
    static std::map<std::pair<int, int64_t>, Tensor> bufs;
    const auto st = c10::hip::getCurrentHIPStream(dev.index());
    std::lock_guard<std::mutex> lk(mu);
    Tensor& b = bufs[{(int)dev.index(), (int64_t)st.id()}];
    if (!b.defined() || (size_t)b.numel() < nbytes)
        b = at::empty({(int64_t)(nbytes + nbytes / 4 + 4096)}, at::TensorOptions().dtype(at::kByte).device(dev));
    return Scratch{b.data_ptr(), (size_t)b.numel(), b};
}

struct Layout {                           // packed valid-cell layout, mask-driven: every listed cell has m = 1
    Tensor cells, row_ptr, cellmap;
    int N = 0, B = 0, L = 0;
};

// the geometry's clip-boundary table, built once per (device, T, L, C)
std::pair<Tensor, Tensor> clip_event_table(const at::Device& dev, int T, int L, int C, bool* built_now = nullptr)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, std::pair<Tensor, Tensor>> tabs;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_tuple((int)dev.index(), T, L, C);
    auto it = tabs.find(key);
    if (built_now) *built_now = it == tabs.end();                 // built on the CURRENT stream: a consumer on another stream has to wait for it
    if (it != tabs.end()) return it->second;
    auto io = at::TensorOptions().dtype(at::kInt).device(dev);
    Tensor counts = at::empty({T}, io);
    SMIN_CK(smin_clip_event_table(cur(), T, L, C, counts.data_ptr<int32_t>(), nullptr, nullptr));
    Tensor offsets = at::zeros({T + 1}, io);
    offsets.slice(0, 1).copy_(at::cumsum(counts, 0).to(at::kInt));
    const int64_t n = std::max<int64_t>(1, offsets[T].item<int64_t>());
    Tensor table = at::empty({n, 2}, io);
    SMIN_CK(smin_clip_event_table(cur(), T, L, C, nullptr, ip(offsets), table.data_ptr()));
    return tabs.emplace(key, std::make_pair(offsets, table)).first->second;
}

Tensor undef() { return Tensor(); }

// device word that smin_build_cells_n sets when a caller-supplied cell count does not match the mask (see csrc/layout.hip)
Tensor layout_status(const at::Device& dev)
{
    static std::mutex mu;
    static std::map<int, Tensor> st;
    std::lock_guard<std::mutex> lk(mu);
    Tensor& t = st[(int)dev.index()];
    if (!t.defined()) t = at::zeros({1}, at::TensorOptions().dtype(at::kInt).device(dev));
    return t;
}

// the two device words smin_step_prologue counts with (zero between launches)
Tensor prologue_words(const at::Device& dev)
{
    static std::mutex mu;
    static std::map<int, Tensor> st;
    std::lock_guard<std::mutex> lk(mu);
    Tensor& t = st[(int)dev.index()];
    if (!t.defined()) t = at::zeros({2}, at::TensorOptions().dtype(at::kLong).device(dev));
    return t;
}

// ---------------------------------------------------------------- gradient exchange inside the node (data parallel)
// With the whole model as one autograd node torch DDP sees every gradient only when the node returns: its all-reduce would start
// after the backward pass.  Instead the node hands each group of finished gradient buffers to the process group itself
// (c10d functional collectives, i.e. RCCL on a HIP device), on the stream that produced them, as soon as they are final -- the
// boundary unit's and moment unit's weights layer by layer, the inputs of the parameter-product kernel before it runs (its outputs
// are linear in them with coefficients that are equal on every rank, so they come out averaged), the word-side and localization
// gradients behind their kernels, the backbone's at the end -- and the main stream joins the collectives before the node returns.
struct GradSyncConfig { std::string group; int world = 1; bool coalesced_avg = false; };
GradSyncConfig& grad_sync_config() { static GradSyncConfig c; return c; }

struct GradSync {
    bool on = false;
    std::vector<Tensor> pending;
    static c10::OperatorHandle op(const char* name) { return c10::Dispatcher::singleton().findSchemaOrThrow(name, ""); }
    // every tensor: a contiguous gradient buffer, handed over exactly once; `on_stream` = the stream its producer ran on
    void reduce(const std::vector<Tensor>& ts_in, HStream on_stream)
    {
        if (!on) return;
        std::vector<Tensor> ts;
        for (auto& t : ts_in) if (t.defined() && t.numel() > 0) { TORCH_CHECK(t.is_contiguous(), "grad sync: non-contiguous gradient buffer"); ts.push_back(t); }
        if (ts.empty()) return;
        const GradSyncConfig& c = grad_sync_config();
        StreamScope sc(on_stream);
        if (c.coalesced_avg) {                                   // RCCL: one grouped launch, averaged by the library
            static auto h = op("_c10d_functional::all_reduce_coalesced_");
            torch::jit::Stack stack;
            stack.emplace_back(ts); stack.emplace_back(std::string("avg")); stack.emplace_back(c.group);
            h.callBoxed(stack);
            for (auto& t : ts) pending.push_back(t);
        } else {                                                  // gloo (tests): per tensor, summed, then scaled
            static auto h = op("_c10d_functional::all_reduce_");
            static auto w = op("_c10d_functional::wait_tensor");
            for (auto& t : ts) {
                torch::jit::Stack stack;
                stack.emplace_back(t); stack.emplace_back(std::string("sum")); stack.emplace_back(c.group);
                h.callBoxed(stack);
                torch::jit::Stack ws; ws.emplace_back(t);
                w.callBoxed(ws);
                t.mul_(1.0 / c.world);
            }
        }
    }
    // `waiter` waits for every collective handed over so far
    void join(HStream waiter)
    {
        if (pending.empty()) return;
        static auto w = op("_c10d_functional::wait_tensor");
        StreamScope sc(waiter);
        for (auto& t : pending) { torch::jit::Stack ws; ws.emplace_back(t); w.callBoxed(ws); }
        pending.clear();
    }
};

// parameter order (modules.py: SMIN._native_params): video encoder 3, LSTM 16, 20 per SMI layer, localization 8
enum { P_VE_W = 0, P_VE_B, P_PE, P_LSTM = 3, P_LAYER0 = 19 };
enum { L_CH_W = 0, L_CH_B, L_WH_W, L_WH_B, L_SH_W, L_SH_B, L_C_W, L_C_B, L_AQ_W, L_AQ_B, L_AK_W, L_AK_B, L_BQ_W, L_BQ_B, L_BK_W, L_BK_B, L_FB_W, L_FB_B, L_FC_W,
       L_FC_B, L_COUNT };


// ---------------------------------------------------------------- autograd nodes (one per reference module body)

// f = ((x W^T + b + pe[t]) * vmask) * f_s   -- VideoEncoder.forward + Backbone's Hadamard product (models.py:25-36, 81-83)
struct VideoFuse : torch::autograd::Function<VideoFuse> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor W, Tensor bias, Tensor pe, Tensor vmask, Tensor fs)
    {
        x = cont(x); W = cont(W); bias = cont(bias); pe = cont(pe); vmask = cont(vmask); fs = cont(fs);
        const int B = i32(x.size(0)), T = i32(x.size(1)), Din = i32(x.size(2)), D = i32(W.size(0));
        Tensor fv = at::empty({B, T, D}, x.options()), f = at::empty({B, T, D}, x.options());
        SMIN_CK(smin_video_encoder_fwd(cur(), fp(x), fp(W), fp(bias), fp(pe), fp(vmask), fp(fs), B, T, Din, D, fpm(fv), fpm(f)));
        ctx->save_for_backward({x, fv, fs, vmask});
        ctx->saved_data["pe_rows"] = pe.size(0);
        return f;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor &x = sv[0], &fv = sv[1], &fs = sv[2], &vmask = sv[3];
        const int B = i32(x.size(0)), T = i32(x.size(1)), Din = i32(x.size(2)), D = i32(fv.size(2));
        const int64_t pe_rows = ctx->saved_data["pe_rows"].toInt();
        Tensor df = cont(g[0]);
        Tensor dW = at::empty({D, Din}, x.options()), dbias = at::empty({D}, x.options());
        Tensor dpe = pe_rows != T ? at::zeros({pe_rows, D}, x.options()) : at::empty({T, D}, x.options());
        Tensor dfs = at::empty_like(fs);
        auto ws = scratch(smin_video_encoder_bwd_workspace_bytes(B, T, Din, D), x.device());
        SMIN_CK(smin_video_encoder_bwd(cur(), fp(df), fp(fv), fp(fs), fp(vmask), fp(x), B, T, Din, D, fpm(dW), fpm(dbias), fpm(dpe), fpm(dfs),
                                       ws.p, ws.n));
        return {undef(), dW, dbias, dpe, undef(), dfs};
    }
};

// one bidirectional LSTM layer over a padded batch with per-sample lengths (models.py:46-58)
struct BiLstmLayer : torch::autograd::Function<BiLstmLayer> {
    static Tensor forward(AutogradContext* ctx, Tensor x, Tensor length, Tensor w_ih_f, Tensor w_hh_f, Tensor b_ih_f, Tensor b_hh_f,
                          Tensor w_ih_r, Tensor w_hh_r, Tensor b_ih_r, Tensor b_hh_r)
    {
        x = cont(x);
        const int B = i32(x.size(0)), Nq = i32(x.size(1)), In = i32(x.size(2)), H = i32(w_hh_f.size(1));
        Tensor Wih = at::cat({w_ih_f, w_ih_r}).contiguous();                           // [8H, In]
        Tensor bias = at::cat({b_ih_f + b_hh_f, b_ih_r + b_hh_r}).contiguous();        // [8H]
        Tensor Whh = at::stack({w_hh_f, w_hh_r}).contiguous();                         // [2, 4H, H]
        Tensor W4 = Whh.view({2, 4, H, H}).permute({0, 3, 2, 1}).contiguous();         // [2, k, u, gate]
        Tensor G = at::empty({B, Nq, 2, 4 * H}, x.options()), Hout = at::empty({B, Nq, 2 * H}, x.options()), Cs = at::empty({B, Nq, 2, H}, x.options());
        SMIN_CK(smin_bilstm_layer_fwd(cur(), fp(x), fp(Wih), fp(bias), fp(W4), ip(length), B, Nq, In, H, fpm(G), fpm(Hout), fpm(Cs)));
        ctx->save_for_backward({x, Hout, G, Cs, Wih, Whh});
        ctx->saved_data["len"] = length;
        return Hout;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor &x = sv[0], &Hout = sv[1], &G = sv[2], &Cs = sv[3], &Wih = sv[4], &Whh = sv[5];
        const Tensor length = ctx->saved_data["len"].toTensor();
        const int B = i32(x.size(0)), Nq = i32(x.size(1)), In = i32(x.size(2)), H = i32(Whh.size(2));
        Tensor dH = cont(g[0]);
        Tensor dX = ctx->needs_input_grad(0) ? at::empty_like(x) : Tensor();
        Tensor dWih = at::empty_like(Wih), dbias = at::empty({8 * H}, x.options()), dWhh = at::empty_like(Whh);
        auto ws = scratch(smin_bilstm_layer_bwd_workspace_bytes(B, Nq, In, H), x.device());
        Tensor WihT = Wih.t().contiguous();
        SMIN_CK(smin_bilstm_layer_bwd(cur(), fp(dH), fp(x), fp(Hout), fp(G), fp(Cs), fp(WihT), fp(Whh), ip(length), B, Nq, In, H,
                                      fpm(dX), fpm(dWih), fpm(dbias), fpm(dWhh), ws.p, ws.n));
        const int64_t H4 = 4 * H;
        Tensor db_f = dbias.slice(0, 0, H4), db_r = dbias.slice(0, H4);
        return {dX, undef(), dWih.slice(0, 0, H4), dWhh[0], db_f, db_f, dWih.slice(0, H4), dWhh[1], db_r, db_r};
    }
};

// ProposalGeneration.forward without f_c: (f_m, f_b) (models.py:115-126; the content stream never forms f_c)
struct ProposalMeans : torch::autograd::Function<ProposalMeans> {
    static variable_list forward(AutogradContext* ctx, Tensor f, Tensor cells, Tensor row_ptr, Tensor cellmap, int64_t N, int64_t T, int64_t L, int64_t C)
    {
        f = cont(f);
        const int B = i32(f.size(0)), D = i32(f.size(2));
        TORCH_CHECK(f.size(1) == T, "ProposalGeneration was built for T=", T, " but got ", f.size(1), " frames");
        Tensor fm = at::empty({N, D}, f.options()), fb = at::empty({B, L, D}, f.options());
        auto ws = scratch((size_t)8 * B * (T + 1) * D, f.device());
        SMIN_CK(smin_proposal_map_fwd(cur(), fp(f), ip(cells), i32(N), B, i32(T), i32(L), i32(C), D, nullptr, fpm(fm), fpm(fb), ws.p, ws.n));
        ctx->saved_data["cells"] = cells; ctx->saved_data["row_ptr"] = row_ptr; ctx->saved_data["cellmap"] = cellmap;
        ctx->saved_data["d"] = std::vector<int64_t>{N, B, T, L, C, D};
        return {fm, fb};
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto d = ctx->saved_data["d"].toIntVector();
        const int N = i32(d[0]), B = i32(d[1]), T = i32(d[2]), L = i32(d[3]), C = i32(d[4]), D = i32(d[5]);
        const Tensor cells = ctx->saved_data["cells"].toTensor(), row_ptr = ctx->saved_data["row_ptr"].toTensor(), cellmap = ctx->saved_data["cellmap"].toTensor();
        Tensor dfm = cont(g[0]), dfb = cont(g[1]);
        const Tensor& ref = dfm.defined() ? dfm : dfb;
        Tensor df = at::empty({B, T, D}, ref.options());
        auto ws = scratch((size_t)4 * B * T * D, df.device());
        auto tab = clip_event_table(df.device(), T, L, C);
        SMIN_CK(smin_proposal_map_bwd(cur(), nullptr, fp(dfm), fp(dfb), ip(cells), ip(row_ptr), ip(cellmap), N, B, T, L, C, D, fpm(df), ws.p, ws.n,
                                      ip(tab.first), tab.second.data_ptr()));
        return {df, undef(), undef(), undef(), undef(), undef(), undef(), undef()};
    }
};

// out[s][n, c] = mean over clip c of cell n of g[b, t, s*W:(s+1)*W] + bias[s*W ..]  (models.py:117, 247 through g = f [Wch_1; ..]^T)
struct ClipWindowMeans : torch::autograd::Function<ClipWindowMeans> {
    static variable_list forward(AutogradContext* ctx, Tensor g, Tensor bias, Tensor cells, Tensor row_ptr, Tensor cellmap, int64_t N, int64_t T,
                                 int64_t L, int64_t C, int64_t nseg)
    {
        g = cont(g); bias = cont(bias);
        const int B = i32(g.size(0)), D = i32(g.size(2)), W = D / i32(nseg);
        Tensor out = at::empty({nseg, N * C, W}, g.options());
        auto ws = scratch((size_t)8 * B * (T + 1) * D, g.device());
        const int nb = bias.defined() ? i32(bias.numel()) : 0;
        SMIN_CK(smin_clip_window_means_fwd(cur(), fp(g), fp(bias), nb, ip(cells), i32(N), B, i32(T), i32(L), i32(C), W, i32(nseg), fpm(out), ws.p, ws.n));
        ctx->saved_data["cells"] = cells; ctx->saved_data["row_ptr"] = row_ptr; ctx->saved_data["cellmap"] = cellmap;
        ctx->saved_data["d"] = std::vector<int64_t>{N, B, T, L, C, W, nseg, nb};
        variable_list outs;
        for (int64_t s = 0; s < nseg; ++s) outs.push_back(out[s]);
        return outs;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto d = ctx->saved_data["d"].toIntVector();
        const int N = i32(d[0]), B = i32(d[1]), T = i32(d[2]), L = i32(d[3]), C = i32(d[4]), W = i32(d[5]), nseg = i32(d[6]), nb = i32(d[7]);
        const Tensor cells = ctx->saved_data["cells"].toTensor(), row_ptr = ctx->saved_data["row_ptr"].toTensor(), cellmap = ctx->saved_data["cellmap"].toTensor();
        at::TensorOptions opt;
        for (auto& t : g) if (t.defined()) { opt = t.options(); break; }
        std::vector<Tensor> douts(nseg);
        std::vector<const float*> ptrs(nseg);
        for (int s = 0; s < nseg; ++s) {
            douts[s] = g[s].defined() ? cont(g[s]) : at::zeros({(int64_t)N * C, W}, opt);
            ptrs[s] = fp(douts[s]);
        }
        Tensor dg = at::empty({B, T, (int64_t)W * nseg}, opt);
        auto ws = scratch((size_t)4 * B * T * W * nseg, dg.device());
        auto tab = clip_event_table(dg.device(), T, L, C);
        SMIN_CK(smin_clip_window_means_bwd(cur(), ptrs.data(), ip(cells), ip(row_ptr), ip(cellmap), N, B, T, L, C, W, nseg, fpm(dg), ws.p, ws.n,
                                           ip(tab.first), tab.second.data_ptr()));
        Tensor dbias;
        if (ctx->needs_input_grad(1) && nb) {
            std::vector<Tensor> parts;
            for (int s = 0; s < nb / W; ++s) parts.push_back(douts[s].sum(0));
            dbias = at::cat(parts);
        }
        return {dg, dbias, undef(), undef(), undef(), undef(), undef(), undef(), undef(), undef()};
    }
};

// the content unit's attention core (models.py:252-267): chat [N*C, dl] -> (cc [N*C, dl], ccmean [N, dl])
struct ContentAttn : torch::autograd::Function<ContentAttn> {
    static variable_list forward(AutogradContext* ctx, Tensor chat, Tensor Mq, Tensor uq, Tensor what, Tensor shat, Tensor qmask, Tensor cells, Tensor row_ptr,
                                 int64_t N, int64_t L, int64_t C, bool want_rows)
    {
        chat = cont(chat); Mq = cont(Mq); uq = cont(uq); what = cont(what); shat = cont(shat); qmask = cont(qmask);
        const int B = i32(what.size(0)), Nq = i32(what.size(1)), dl = i32(what.size(2));
        Tensor cc = at::empty({want_rows ? N * C : 0, dl}, chat.options()), ccmean = at::empty({N, dl}, chat.options());
        SMIN_CK(smin_content_attn_fwd(cur(), fp(chat), ip(cells), ip(row_ptr), i32(N), B, i32(L), i32(C), dl, Nq, fp(Mq), fp(uq), fp(what), fp(shat), fp(qmask),
                                      want_rows ? fpm(cc) : nullptr, fpm(ccmean)));
        ctx->save_for_backward({chat, Mq, uq, what, shat, qmask});
        ctx->saved_data["cells"] = cells; ctx->saved_data["row_ptr"] = row_ptr;
        ctx->saved_data["d"] = std::vector<int64_t>{N, L, C, want_rows};
        if (!want_rows) ctx->mark_non_differentiable({cc});
        return {cc, ccmean};
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor &chat = sv[0], &Mq = sv[1], &uq = sv[2], &what = sv[3], &shat = sv[4], &qmask = sv[5];
        auto d = ctx->saved_data["d"].toIntVector();
        const int N = i32(d[0]), L = i32(d[1]), C = i32(d[2]);
        const bool want_rows = d[3] != 0;
        const Tensor cells = ctx->saved_data["cells"].toTensor(), row_ptr = ctx->saved_data["row_ptr"].toTensor();
        const int B = i32(what.size(0)), Nq = i32(what.size(1)), dl = i32(what.size(2));
        Tensor dcc = want_rows ? cont(g[0]) : Tensor(), dccmean = cont(g[1]);
        if (!dcc.defined() && !dccmean.defined()) dccmean = at::zeros({N, dl}, chat.options());
        Tensor dchat = at::empty_like(chat), dMq = at::empty_like(Mq), duq = at::empty_like(uq), dwhat = at::empty_like(what), dshat = at::empty_like(shat);
        if (N == 0) { dMq.zero_(); duq.zero_(); dwhat.zero_(); dshat.zero_(); }
        else {
            auto ws = scratch(smin_content_attn_bwd_workspace_bytes(N, B, C, dl), chat.device());
            SMIN_CK(smin_content_attn_bwd(cur(), fp(dcc), fp(dccmean), fp(chat), ip(cells), ip(row_ptr), N, B, L, C, dl, Nq, fp(Mq), fp(uq), fp(what), fp(shat),
                                          fp(qmask), fpm(dchat), fpm(dMq), fpm(duq), fpm(dwhat), fpm(dshat), ws.p, ws.n));
        }
        return {dchat, dMq, duq, dwhat, dshat, undef(), undef(), undef(), undef(), undef(), undef(), undef()};
    }
};

// y[r] = [x_0[r] | x_1[r] | ..] W^T + bias + add_rows[r] + add_cells[r / C]
struct LinearRows : torch::autograd::Function<LinearRows> {
    // (optional inputs are std::optional: an undefined Tensor is not a legal autograd input)
    static Tensor forward(AutogradContext* ctx, Tensor W, std::optional<Tensor> bias_, std::optional<Tensor> add_rows_, std::optional<Tensor> add_cells_, int64_t C,
                          at::TensorList xs_in)          // (a std::vector<Tensor> argument would be taken for a non-tensor)
    {
        std::vector<Tensor> xs(xs_in.begin(), xs_in.end());
        W = cont(W);
        Tensor bias = bias_ ? cont(*bias_) : Tensor(), add_rows = add_rows_ ? cont(*add_rows_) : Tensor(), add_cells = add_cells_ ? cont(*add_cells_) : Tensor();
        for (auto& x : xs) x = cont(x);
        const int R = i32(xs[0].size(0)), K = i32(xs[0].size(1)), O = i32(W.size(0)), nseg = i32(xs.size());
        Tensor y = at::empty({R, O}, xs[0].options());
        std::vector<const float*> ptrs;
        for (auto& x : xs) ptrs.push_back(fp(x));
        SMIN_CK(smin_linear_rows_fwd(cur(), ptrs.data(), nseg, fp(W), fp(bias), fp(add_rows), fp(add_cells), i32(C), R, O, K, fpm(y)));
        variable_list save{W};
        for (auto& x : xs) save.push_back(x);
        ctx->save_for_backward(save);
        // AutogradContext::needs_input_grad counts graph edges, i.e. only the tensor inputs that are present
        int64_t e = 1;
        const int64_t e_bias = bias_ ? e++ : -1, e_rows = add_rows_ ? e++ : -1, e_cells = add_cells_ ? e++ : -1;
        ctx->saved_data["d"] = std::vector<int64_t>{C, e_bias, e_rows, e_cells, e};
        return y;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor& W = sv[0];
        const int nseg = i32(sv.size()) - 1;
        auto dd = ctx->saved_data["d"].toIntVector();
        const int64_t C = dd[0], e_bias = dd[1], e_rows = dd[2], e_cells = dd[3], e_x = dd[4];
        auto need = [&](int64_t e) { return e >= 0 && ctx->needs_input_grad((size_t)e); };
        Tensor dy = cont(g[0]);
        const int R = i32(sv[1].size(0)), K = i32(sv[1].size(1)), O = i32(W.size(0));
        bool want_dx = false;
        for (int s = 0; s < nseg; ++s) want_dx = want_dx || need(e_x + s);
        std::vector<Tensor> dxs;
        std::vector<const float*> xp;
        std::vector<float*> dxp;
        for (int s = 0; s < nseg; ++s) {
            xp.push_back(fp(sv[1 + s]));
            if (want_dx) { dxs.push_back(at::empty_like(sv[1 + s])); dxp.push_back(fpm(dxs.back())); }
        }
        Tensor dW = at::empty_like(W);
        Tensor dbias = need(e_bias) ? at::empty({O}, dy.options()) : Tensor();
        auto ws = scratch(smin_linear_rows_bwd_workspace_bytes(R, O, nseg * K), dy.device());
        Tensor WT = W.t().contiguous();
        SMIN_CK(smin_linear_rows_bwd(cur(), fp(dy), xp.data(), nseg, fp(WT), R, O, K, want_dx ? dxp.data() : nullptr, fpm(dW), fpm(dbias), ws.p, ws.n));
        Tensor dcells;
        if (need(e_cells)) {
            if (C == 1) dcells = dy;
            else {
                dcells = at::empty({R / C, O}, dy.options());
                SMIN_CK(smin_group_sum(cur(), fp(dy), i32(R / C), i32(C), O, fpm(dcells)));
            }
        }
        variable_list out{dW, dbias, need(e_rows) ? dy : Tensor(), dcells, undef()};
        for (int s = 0; s < nseg; ++s) out.push_back(want_dx ? dxs[s] : Tensor());
        return out;
    }
};

// word-side operands of every layer's content attention in one launch (models.py:249-251, 209-211; csrc/word_prep.hip):
// params = 8 per layer (linear_w_hat, linear_s_hat, attn_layer.W_k, attn_layer.W_q: weight, bias); returns (what, shat, Mq, uq) per layer
struct WordPrep : torch::autograd::Function<WordPrep> {
    static variable_list forward(AutogradContext* ctx, Tensor fw, Tensor fs, Tensor qmask, at::TensorList params_in)
    {
        fw = cont(fw); fs = cont(fs); qmask = cont(qmask);
        std::vector<Tensor> params;
        std::vector<const float*> pp;
        for (const Tensor& p : params_in) { params.push_back(cont(p)); pp.push_back(fp(params.back())); }
        const int nl = i32(params.size() / 8), B = i32(fw.size(0)), Nq = i32(fw.size(1)), D = i32(fw.size(2)), dl = i32(params[0].size(0));
        Tensor what = at::empty({nl, B, Nq, dl}, fw.options()), kb = at::empty({nl, B, Nq, dl}, fw.options()), Mq = at::empty({nl, B, Nq, dl}, fw.options());
        Tensor shat = at::empty({nl, B, dl}, fw.options()), uq = at::empty({nl, B, Nq}, fw.options());
        SMIN_CK(smin_word_prep_fwd(cur(), fp(fw), fp(fs), fp(qmask), pp.data(), nl, B, Nq, D, dl, fpm(what), fpm(shat), fpm(kb), fpm(Mq), fpm(uq)));
        variable_list save{fw, fs, qmask, what, kb};
        for (auto& p : params) save.push_back(p);
        ctx->save_for_backward(save);
        variable_list outs;
        for (int k = 0; k < nl; ++k) { outs.push_back(what[k]); outs.push_back(shat[k]); outs.push_back(Mq[k]); outs.push_back(uq[k]); }
        return outs;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor &fw = sv[0], &fs = sv[1], &qmask = sv[2], &what = sv[3], &kb = sv[4];
        const int nl = i32((sv.size() - 5) / 8), B = i32(fw.size(0)), Nq = i32(fw.size(1)), D = i32(fw.size(2)), dl = i32(sv[5].size(0));
        std::vector<Tensor> keep;
        std::vector<const float*> gp[4], pp;
        for (int k = 0; k < nl; ++k)
            for (int j = 0; j < 4; ++j) {
                const Tensor& t = g[4 * k + j];
                if (t.defined()) { keep.push_back(cont(t)); gp[j].push_back(fp(keep.back())); } else gp[j].push_back(nullptr);
            }
        Tensor dfw = at::empty_like(fw), dfs = at::empty_like(fs);
        std::vector<Tensor> dparams;
        std::vector<float*> dp;
        for (int i = 0; i < nl * 8; ++i) { pp.push_back(fp(sv[5 + i])); dparams.push_back(at::empty_like(sv[5 + i])); dp.push_back(fpm(dparams.back())); }
        auto ws = scratch(smin_word_prep_bwd_workspace_bytes(nl, B, Nq, D, dl), fw.device());
        SMIN_CK(smin_word_prep_bwd(cur(), gp[0].data(), gp[1].data(), gp[2].data(), gp[3].data(), fp(fw), fp(fs), fp(qmask), fp(what), fp(kb), pp.data(), nl, B, Nq, D, dl,
                                   fpm(dfw), fpm(dfs), dp.data(), ws.p, ws.n));
        variable_list out{dfw, dfs, undef()};
        for (auto& t : dparams) out.push_back(t);
        return out;
    }
};

// hbar = sigmoid(fm * fs) * fm (models.py:191, 272-274), n_hbar views of it followed by n_res views of fm: every
// consumer gets its own view so that the one backward kernel sums their gradients
struct Gate : torch::autograd::Function<Gate> {
    static variable_list forward(AutogradContext* ctx, Tensor fm, Tensor fs, Tensor cells, Tensor row_ptr, int64_t B, int64_t L, int64_t n_hbar, int64_t n_res)
    {
        fm = cont(fm); fs = cont(fs);
        const int N = i32(fm.size(0)), D = i32(fm.size(1));
        Tensor hbar = at::empty_like(fm);
        SMIN_CK(smin_gate_fwd(cur(), fp(fm), fp(fs), ip(cells), N, D, fpm(hbar)));
        ctx->save_for_backward({fm, fs});
        ctx->saved_data["row_ptr"] = row_ptr;
        ctx->saved_data["d"] = std::vector<int64_t>{B, L, n_hbar};
        variable_list outs{hbar};
        for (int64_t k = 1; k < n_hbar; ++k) outs.push_back(at::alias(hbar));
        for (int64_t k = 0; k < n_res; ++k) outs.push_back(at::alias(fm));
        return outs;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor &fm = sv[0], &fs = sv[1];
        auto d = ctx->saved_data["d"].toIntVector();
        const int B = i32(d[0]), L = i32(d[1]), n_hbar = i32(d[2]);
        const Tensor row_ptr = ctx->saved_data["row_ptr"].toTensor();
        const int N = i32(fm.size(0)), D = i32(fm.size(1));
        std::vector<Tensor> keep;
        std::vector<const float*> dh, dr;
        for (int k = 0; k < (int)g.size(); ++k) {
            if (!g[k].defined()) continue;
            keep.push_back(cont(g[k]));
            (k < n_hbar ? dh : dr).push_back(fp(keep.back()));
        }
        if (dh.empty()) { keep.push_back(at::zeros_like(fm)); dh.push_back(fp(keep.back())); }
        Tensor dfm = at::empty_like(fm), dfs = at::empty_like(fs);
        auto ws = scratch((size_t)4 * B * 512 * D + 4096, fm.device());
        SMIN_CK(smin_gate_bwd(cur(), dh.data(), i32(dh.size()), dr.empty() ? nullptr : dr.data(), i32(dr.size()), fp(fm), fp(fs), ip(row_ptr), N, B, L, D,
                              fpm(dfm), fpm(dfs), ws.p, ws.n, nullptr, nullptr, nullptr));
        return {dfm, dfs, undef(), undef(), undef(), undef(), undef(), undef()};
    }
};

// BoundaryUnit.forward with its word attention (models.py:137-196)
struct BoundaryUnitFn : torch::autograd::Function<BoundaryUnitFn> {
    static Tensor forward(AutogradContext* ctx, Tensor fb, Tensor fw, Tensor fs, Tensor hbar, Tensor Wq, Tensor bq, Tensor Wk, Tensor bk, Tensor qmask, Tensor lmask,
                          Tensor cells, Tensor row_ptr, int64_t N)
    {
        fb = cont(fb); fw = cont(fw); fs = cont(fs); hbar = cont(hbar); Wq = cont(Wq); bq = cont(bq); Wk = cont(Wk); bk = cont(bk); qmask = cont(qmask); lmask = cont(lmask);
        const int B = i32(fb.size(0)), L = i32(fb.size(1)), D = i32(fb.size(2)), Nq = i32(fw.size(1));
        Tensor out = at::empty_like(fb), Qb = at::empty_like(fb), baq = at::empty_like(fb), bqv = at::empty_like(fb), Kb = at::empty_like(fw);
        Tensor P = at::empty({B, L, Nq}, fb.options()), A = at::empty({B, L, L}, fb.options());
        SMIN_CK(smin_boundary_unit_fwd(cur(), fp(fb), fp(fw), fp(fs), fp(hbar), ip(cells), ip(row_ptr), i32(N), B, L, Nq, D, fp(Wq), fp(bq), fp(Wk), fp(bk),
                                       fp(qmask), fp(lmask), fpm(out), fpm(Qb), fpm(Kb), fpm(P), fpm(baq), fpm(bqv), fpm(A)));
        ctx->save_for_backward({fb, fw, fs, hbar, Wq, Wk, qmask, lmask, Qb, Kb, P, baq, bqv, A});
        ctx->saved_data["cells"] = cells; ctx->saved_data["row_ptr"] = row_ptr; ctx->saved_data["N"] = N;
        return out;
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor &fb = sv[0], &fw = sv[1], &fs = sv[2], &hbar = sv[3], &Wq = sv[4], &Wk = sv[5], &qmask = sv[6], &lmask = sv[7], &Qb = sv[8], &Kb = sv[9],
                     &P = sv[10], &baq = sv[11], &bqv = sv[12], &A = sv[13];
        const Tensor cells = ctx->saved_data["cells"].toTensor(), row_ptr = ctx->saved_data["row_ptr"].toTensor();
        const int N = i32(ctx->saved_data["N"].toInt());
        const int B = i32(fb.size(0)), L = i32(fb.size(1)), D = i32(fb.size(2)), Nq = i32(fw.size(1));
        Tensor dout = cont(g[0]);
        Tensor WqT = Wq.t().contiguous(), WkT = Wk.t().contiguous();
        Tensor dfb = at::empty_like(fb), dfw = at::empty_like(fw), dfs = at::empty_like(fs), dhbar = at::empty_like(hbar);
        Tensor dWq = at::empty_like(Wq), dbq = at::empty({D}, fb.options()), dWk = at::empty_like(Wk), dbk = at::empty({D}, fb.options());
        const size_t nbytes = 4 * ((size_t)2 * B * L * L + (size_t)3 * B * L * D + (size_t)B * L * Nq + (size_t)B * Nq * D + (size_t)2 * 64 * ((size_t)D * D + D)) + 4096;
        auto ws = scratch(nbytes, fb.device());
        SMIN_CK(smin_boundary_unit_bwd(cur(), fp(dout), fp(fb), fp(fw), fp(fs), fp(hbar), ip(cells), ip(row_ptr), N, B, L, Nq, D, fp(WqT), fp(WkT), fp(qmask), fp(lmask),
                                       fp(Qb), fp(Kb), fp(P), fp(baq), fp(bqv), fp(A), fpm(dfb), fpm(dfw), fpm(dfs), fpm(dhbar), fpm(dWq), fpm(dbq), fpm(dWk), fpm(dbk),
                                       ws.p, ws.n));
        return {dfb, dfw, dfs, dhbar, dWq, dbq, dWk, dbk, undef(), undef(), undef(), undef(), undef()};
    }
};

// MomentUnit.forward (models.py:288-303); Wcat = [conv_fb.W | conv_fc.W] (D, 2D).  Returns (mu, view of fcmean)
struct MomentUnitFn : torch::autograd::Function<MomentUnitFn> {
    static variable_list forward(AutogradContext* ctx, Tensor fcmean, Tensor fm, Tensor fb, Tensor Wcat, Tensor bcat, Tensor cells, Tensor row_ptr, Tensor cellmap)
    {
        fcmean = cont(fcmean); fm = cont(fm); fb = cont(fb); Wcat = cont(Wcat); bcat = cont(bcat);
        const int N = i32(fm.size(0)), D = i32(fm.size(1)), B = i32(fb.size(0)), L = i32(fb.size(1));
        Tensor mu = at::empty_like(fm), x1 = at::empty_like(fm);           // x1 = f_b[i] * f_b[j], kept for the weight gradient
        SMIN_CK(smin_pair_product(cur(), fp(fb), ip(cells), N, L, D, fpm(x1)));
        SMIN_CK(smin_moment_unit_fwd(cur(), fp(fcmean), fp(fm), fp(fb), ip(cells), N, B, L, D, fp(Wcat), fp(bcat), fpm(mu), fp(x1)));
        ctx->save_for_backward({fcmean, fb, Wcat, x1});
        ctx->saved_data["cells"] = cells; ctx->saved_data["row_ptr"] = row_ptr; ctx->saved_data["cellmap"] = cellmap;
        return {mu, at::alias(fcmean)};
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor &fcmean = sv[0], &fb = sv[1], &Wcat = sv[2], &x1 = sv[3];
        const Tensor cells = ctx->saved_data["cells"].toTensor(), row_ptr = ctx->saved_data["row_ptr"].toTensor(), cellmap = ctx->saved_data["cellmap"].toTensor();
        const int N = i32(fcmean.size(0)), D = i32(fcmean.size(1)), B = i32(fb.size(0)), L = i32(fb.size(1));
        Tensor dmu = cont(g[0]), dacc = cont(g[1]);
        if (!dmu.defined()) dmu = at::zeros_like(fcmean);
        Tensor WcatT = Wcat.t().contiguous();
        Tensor dfcmean = at::empty_like(fcmean), dfb = at::empty_like(fb), dWcat = at::empty_like(Wcat), dbcat = at::empty({D}, fb.options());
        auto ws = scratch(smin_workspace_bytes(N, B, 4, D, 4, 1), fb.device());
        SMIN_CK(smin_moment_unit_bwd(cur(), fp(dmu), fp(fcmean), fp(fb), ip(cells), ip(row_ptr), ip(cellmap), N, B, L, D, fp(WcatT), fpm(dfcmean), fpm(dfb), fpm(dWcat),
                                     fpm(dbcat), ws.p, ws.n, 1, fp(dacc), fp(x1), nullptr));
        return {dfcmean, dmu, dfb, dWcat, dbcat, undef(), undef(), undef()};
    }
};

// Localization.forward (models.py:335-344): pm (B, L, L) dense, psea (3, B, L)
struct ScoreMap : torch::autograd::Function<ScoreMap> {
    static variable_list forward(AutogradContext* ctx, Tensor fm, Tensor fb, Tensor wm, Tensor bm, Tensor wb, Tensor bb, Tensor lmask, Tensor cells)
    {
        fm = cont(fm); fb = cont(fb); wm = cont(wm); bm = cont(bm); wb = cont(wb); bb = cont(bb); lmask = cont(lmask);
        const int N = i32(fm.size(0)), D = i32(fm.size(1)), B = i32(fb.size(0)), L = i32(fb.size(1));
        Tensor pm = at::empty({B, L, L}, fm.options()), psea = at::empty({3, B, L}, fm.options());
        SMIN_CK(smin_score_map_fwd(cur(), fp(fm), fp(fb), ip(cells), N, B, L, D, fp(wm), fp(bm), fp(wb), fp(bb), fp(lmask), fpm(pm), fpm(psea)));
        ctx->save_for_backward({fm, fb, wm, wb, lmask, pm, psea});
        ctx->saved_data["cells"] = cells;
        return {pm, psea};
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor &fm = sv[0], &fb = sv[1], &wm = sv[2], &wb = sv[3], &lmask = sv[4], &pm = sv[5], &psea = sv[6];
        const Tensor cells = ctx->saved_data["cells"].toTensor();
        const int N = i32(fm.size(0)), D = i32(fm.size(1)), B = i32(fb.size(0)), L = i32(fb.size(1));
        Tensor dpm = g[0].defined() ? cont(g[0]) : at::zeros_like(pm), dpsea = g[1].defined() ? cont(g[1]) : at::zeros_like(psea);
        Tensor dfm = at::empty_like(fm), dfb = at::empty_like(fb), dwm = at::empty_like(wm), dbm = at::empty({1}, fm.options()), dwb = at::empty_like(wb),
               dbb = at::empty({3}, fm.options());
        auto ws = scratch(smin_workspace_bytes(N, B, 4, D, 4, 1), fm.device());
        SMIN_CK(smin_score_map_bwd(cur(), fp(dpm), fp(dpsea), fp(pm), fp(psea), fp(fm), fp(fb), ip(cells), N, B, L, D, fp(wm), fp(wb), fp(lmask), fpm(dfm), fpm(dfb),
                                   fpm(dwm), fpm(dbm), fpm(dwb), fpm(dbb), ws.p, ws.n));
        return {dfm, dfb, dwm, dbm, dwb, dbb, undef(), undef()};
    }
};

// restated loss of the reference's train loop (main.py:89-116): one forward and one backward kernel
struct LossNode : torch::autograd::Function<LossNode> {
    static Tensor forward(AutogradContext* ctx, Tensor pm, Tensor ps, Tensor pe, Tensor pa, Tensor ym, Tensor sm, Tensor mm, Tensor ys, Tensor ss, Tensor ye, Tensor se,
                          Tensor ya, Tensor lm)
    {
        auto f = [](const Tensor& t) { return cont(fl(t)); };
        auto b = [](const Tensor& t) { return cont((t.scalar_type() == at::kBool || t.scalar_type() == at::kByte) ? t : t.ne(0)); };
        pm = f(pm); ps = f(ps); pe = f(pe); pa = f(pa); sm = f(sm); ss = f(ss); se = f(se);
        ym = b(ym); mm = b(mm); ys = b(ys); ye = b(ye); ya = b(ya); lm = b(lm);
        const int B = i32(ps.size(0)), L = i32(ps.size(1));
        Tensor loss = at::empty({1}, pm.options()), part = at::empty({B, 6}, pm.options());
        auto u8 = [](const Tensor& t) { return static_cast<const uint8_t*>(t.const_data_ptr()); };
        SMIN_CK(smin_loss_fwd(cur(), fp(pm), u8(ym), fp(sm), u8(mm), fp(ps), u8(ys), fp(ss), fp(pe), u8(ye), fp(se), fp(pa), u8(ya), u8(lm), B, L, fpm(loss), fpm(part)));
        ctx->save_for_backward({pm, ps, pe, pa, ym, sm, mm, ys, ss, ye, se, ya, lm, part});
        return loss.reshape({});
    }
    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto sv = ctx->get_saved_variables();
        const Tensor &pm = sv[0], &ps = sv[1], &pe = sv[2], &pa = sv[3], &ym = sv[4], &sm = sv[5], &mm = sv[6], &ys = sv[7], &ss = sv[8], &ye = sv[9], &se = sv[10],
                     &ya = sv[11], &lm = sv[12], &part = sv[13];
        const int B = i32(ps.size(0)), L = i32(ps.size(1));
        Tensor dloss = cont(fl(g[0].reshape({1})));
        // the three score gradients are rows of one buffer: the model's node takes them as they are (no gather)
        Tensor dpm = at::empty_like(pm), d3 = at::empty({3, (int64_t)B, (int64_t)L}, ps.options());
        Tensor dps = d3[0], dpe = d3[1], dpa = d3[2];
        auto u8 = [](const Tensor& t) { return static_cast<const uint8_t*>(t.const_data_ptr()); };
        SMIN_CK(smin_loss_bwd(cur(), fp(dloss), fp(part), fp(pm), u8(ym), fp(sm), u8(mm), fp(ps), u8(ys), fp(ss), fp(pe), u8(ye), fp(se), fp(pa), u8(ya), u8(lm), B, L,
                              fpm(dpm), fpm(dps), fpm(dpe), fpm(dpa)));
        variable_list out{dpm, dps, dpe, dpa};
        for (int k = 0; k < 9; ++k) out.push_back(undef());
        return out;
    }
};


// ---------------------------------------------------------------- the fused core
// ProposalGeneration + every SMI layer + Localization (models.py:101-126, 306-344) as ONE autograd node: forward and backward
// are straight-line sequences of the C entry points with hand-placed stream forks.  What the node-per-module graph above pays
// and this does not: ~60 engine node visits, the engine's out-of-place sums for every tensor with several consumers
// (f_s, f_w, the boundary features, the parameters shared between layers -- ~45 tiny launches per step), a transposed copy per
// weight per backward call (one batched launch here) and the autograd bookkeeping of the parameter products.
struct LayerState {
    Tensor fm, hbar, fb, bu, Qb, Kb, P, baq, bqv, A, Hs, chat, cc, ccmean, cum, x1, consts, Wcat;
    std::vector<Tensor> Pcat;
};
struct LstmState { Tensor x, Hout, G, Cs, Wih, Whh; };
struct CoreState {
    Tensor vx, fv, vmaskf, len32, last, f, fw, fs, qmf, lmf, cells, row_ptr, cellmap, Wch_all, what, kb, Mq, uq, shat, pm, psea, fm_out, wb;
    LstmState lstm[2];
    std::vector<LayerState> layer;
};
template <class F>
void visit_state(CoreState& s, F&& fn)
{
    for (Tensor* t : {&s.vx, &s.fv, &s.vmaskf, &s.len32, &s.last, &s.f, &s.fw, &s.fs, &s.qmf, &s.lmf, &s.cells, &s.row_ptr, &s.cellmap, &s.Wch_all, &s.what, &s.kb, &s.Mq,
                      &s.uq, &s.shat, &s.pm, &s.psea, &s.fm_out, &s.wb})
        fn(*t);
    for (auto& l : s.lstm)
        for (Tensor* t : {&l.x, &l.Hout, &l.G, &l.Cs, &l.Wih, &l.Whh}) fn(*t);
    for (auto& l : s.layer) {
        for (Tensor* t : {&l.fm, &l.hbar, &l.fb, &l.bu, &l.Qb, &l.Kb, &l.P, &l.baq, &l.bqv, &l.A, &l.Hs, &l.chat, &l.cc, &l.ccmean, &l.cum, &l.x1, &l.consts, &l.Wcat})
            fn(*t);
        for (auto& p : l.Pcat) fn(p);
    }
}
void size_state(CoreState& s, int64_t nl)
{
    s.layer.resize(nl);
    for (int64_t k = 0; k < nl; ++k) s.layer[k].Pcat.resize((k + 3) / 4);
}

// W^T of every listed matrix, one launch per 32 matrices
std::vector<Tensor> transpose_all(const std::vector<Tensor>& ws)
{
    std::vector<Tensor> out;
    for (size_t lo = 0; lo < ws.size(); lo += SMIN_BATCH_MAX) {
        const size_t n = std::min(ws.size() - lo, (size_t)SMIN_BATCH_MAX);
        const float* src[SMIN_BATCH_MAX]; float* dst[SMIN_BATCH_MAX]; int32_t rows[SMIN_BATCH_MAX], cols[SMIN_BATCH_MAX];
        for (size_t m = 0; m < n; ++m) {
            const Tensor& w = ws[lo + m];
            TORCH_CHECK(w.dim() == 2 && w.is_contiguous(), "transpose_all: contiguous matrices only");
            out.push_back(at::empty({w.size(1), w.size(0)}, w.options()));
            src[m] = fp(w); dst[m] = fpm(out.back()); rows[m] = i32(w.size(0)); cols[m] = i32(w.size(1));
        }
        SMIN_CK(smin_transpose_batch(cur(), src, dst, rows, cols, i32(n)));
    }
    return out;
}
Tensor sum_list(const std::vector<Tensor>& ts)
{
    TORCH_CHECK(!ts.empty() && ts.size() <= SMIN_BATCH_MAX, "sum_list: 1..", SMIN_BATCH_MAX, " tensors");
    if (ts.size() == 1) return ts[0];
    const float* p[SMIN_BATCH_MAX];
    for (size_t k = 0; k < ts.size(); ++k) { TORCH_CHECK(ts[k].is_contiguous() && ts[k].numel() == ts[0].numel(), "sum_list: shapes differ"); p[k] = fp(ts[k]); }
    Tensor out = at::empty_like(ts[0]);
    SMIN_CK(smin_sum_lists(cur(), p, i32(ts.size()), (size_t)ts[0].numel(), fpm(out)));
    return out;
}

struct SminCore : torch::autograd::Function<SminCore> {
    enum { F_OVERLAP_BOUNDARY = 1, F_OVERLAP_PREP = 2, F_ASYNC_WEIGHTS = 4, F_BF16_OPERANDS = 8, F_GRAD_SYNC = 16, F_NO_TAIL_SPLIT = 32 };
    enum { N_FIXED = 13 };          // forward arguments ahead of the parameter list (tensors and scalars alike take one gradient slot)

    static variable_list forward(AutogradContext* ctx, Tensor video_features, Tensor video_mask, Tensor query_features, Tensor query_mask, Tensor length_mask,
                                 Tensor moment_mask, int64_t T, int64_t L, int64_t C, int64_t nl, int64_t maxq, int64_t H, int64_t flags, at::TensorList prm_in)
    {
        std::vector<Tensor> all;
        for (const Tensor& p : prm_in) all.push_back(cont(p));
        std::vector<Tensor> prm(all.begin() + P_LAYER0, all.end());                // the SMI layers' and the localization head's parameters
        const at::Device dev = video_features.device();
        const auto opt = video_features.options();
        const int64_t Bq = video_features.size(0), Tn = video_features.size(1), Nq_in = query_features.size(1);
        TORCH_CHECK(Tn == T, "ProposalGeneration was built for T=", T, " but got ", Tn, " frames");
        const int B = i32(Bq), D = i32(all[P_VE_W].size(0)), Nq = i32(maxq), dl = i32(prm[L_CH_W].size(0)), Li = i32(L), Ci = i32(C), Ti = i32(T);
        auto lp = [&](int64_t k, int which) -> const Tensor& { return prm[k * L_COUNT + which]; };
        const Tensor* loc = &prm[nl * L_COUNT];
        HStream curs = c10::hip::getCurrentHIPStream(dev.index());
        HStream side = (flags & F_OVERLAP_BOUNDARY) ? side_stream(dev.index()) : curs;
        HStream prep = (flags & F_OVERLAP_PREP) ? side : curs;
        CoreState st;
        size_state(st, nl);

        // ---- parameter-only work (weight products, constants, concatenations: ~25 tiny launches) on the second stream from the first
        // moment of the step, beside the LSTM recurrence that opens the critical path; the main stream waits for it before the proposal map
        auto mark = [](HStream on) { hipEvent_t e = next_event(); TORCH_CHECK(hipEventRecord(e, on.stream()) == hipSuccess, "hipEventRecord failed"); return e; };
        auto await = [](HStream waiter, hipEvent_t e) { TORCH_CHECK(hipStreamWaitEvent(waiter.stream(), e, 0) == hipSuccess, "hipStreamWaitEvent failed"); };
        std::vector<Tensor> bcat(nl);
        Tensor bb;
        hipEvent_t products_ready;
        const bool prep_kernel = D % 32 == 0 && D <= 1056 && dl % 32 == 0 && nl <= 8;   // limits of csrc/param_prep.hip (else: torch calls)
        wait_stream(prep, curs);                                                    // (the optimizer's update of the parameters)
        {
            StreamScope sc(prep);
            if (prep_kernel) {
                Tensor consts_all = at::empty({nl, dl}, opt), Wcat_all = at::empty({nl, D, 2 * D}, opt), bcat_all = at::empty({nl, D}, opt);
                st.Wch_all = at::empty({nl * dl, D}, opt);
                std::vector<const float*> pp;
                std::vector<float*> pc(nl * 2, nullptr);
                for (int64_t k = 0; k < nl; ++k) {
                    for (int which : {L_CH_W, L_CH_B, L_C_W, L_C_B, L_FB_W, L_FB_B, L_FC_W, L_FC_B}) pp.push_back(fp(lp(k, which)));
                    LayerState& ls = st.layer[k];
                    for (int64_t lo = 0; lo < k; lo += 4) {
                        ls.Pcat[lo / 4] = at::empty({dl, std::min<int64_t>(4, k - lo) * dl}, opt);
                        pc[k * 2 + lo / 4] = fpm(ls.Pcat[lo / 4]);
                    }
                    ls.consts = consts_all[k]; ls.Wcat = Wcat_all[k]; bcat[k] = bcat_all[k];
                }
                SMIN_CK(smin_param_prep_fwd(cur(), pp.data(), i32(nl), D, dl, pc.data(), fpm(consts_all), fpm(Wcat_all), fpm(bcat_all), fpm(st.Wch_all)));
            } else {
                Tensor bsum;
                std::vector<Tensor> wch;
                for (int64_t k = 0; k < nl; ++k) {
                    LayerState& ls = st.layer[k];
                    ls.consts = bsum.defined() ? lp(k, L_CH_B) + at::mv(lp(k, L_CH_W), bsum) : lp(k, L_CH_B);
                    bsum = bsum.defined() ? bsum + lp(k, L_C_B) : lp(k, L_C_B);
                    wch.push_back(lp(k, L_CH_W));
                    for (int64_t lo = 0; lo < k; lo += 4) {
                        std::vector<Tensor> parts;
                        for (int64_t l = lo; l < std::min(lo + 4, k); ++l) parts.push_back(at::matmul(lp(k, L_CH_W), lp(l, L_C_W)));
                        ls.Pcat[lo / 4] = parts.size() == 1 ? parts[0] : at::cat(parts, 1);
                    }
                    ls.Wcat = at::cat({lp(k, L_FB_W).view({D, D}), lp(k, L_FC_W).view({D, D})}, 1);
                    bcat[k] = lp(k, L_FB_B) + lp(k, L_FC_B);
                }
                st.Wch_all = nl == 1 ? wch[0] : at::cat(wch);
            }
            products_ready = mark(prep);
        }
        // ---- masks as fp32, query lengths, the cell count, the boundary heads' parameters side by side: one launch (csrc/layout.hip);
        // as torch calls (masks that are not one byte per element) eight launches in front of the query encoder
        const int64_t n_known = (flags >> 16) - 1;
        Tensor qm = query_mask.reshape({Bq, -1});
        auto bytes = [](const Tensor& t) { return t.element_size() == 1 && t.is_contiguous(); };
        const bool fast_prologue = bytes(video_mask) && bytes(qm) && bytes(length_mask) && bytes(moment_mask) && qm.size(1) == maxq && video_mask.numel() == Bq * Tn &&
                                   length_mask.numel() == Bq * L && moment_mask.numel() == Bq * L * L;
        Tensor mm, host_n, qmf, lmf;
        if (fast_prologue) {
            st.wb = at::empty({3, (int64_t)D}, opt); bb = at::empty({3}, opt);
            st.len32 = at::empty({Bq}, opt.dtype(at::kInt));
            qmf = at::empty({Bq, maxq}, opt); lmf = at::empty({Bq, L}, opt); st.vmaskf = at::empty({Bq * Tn}, opt);
            Tensor count = at::empty({1}, opt.dtype(at::kLong));
            const float* w3[3] = {fp(loc[2]), fp(loc[4]), fp(loc[6])};
            const float* b3[3] = {fp(loc[3]), fp(loc[5]), fp(loc[7])};
            auto u8 = [](const Tensor& t) { return static_cast<const uint8_t*>(t.const_data_ptr()); };
            SMIN_CK(smin_step_prologue(cur(), u8(qm), u8(video_mask), u8(length_mask), u8(moment_mask), w3, b3, B, Nq, Ti, Li, D, st.len32.data_ptr<int32_t>(), fpm(qmf),
                                       fpm(st.vmaskf), fpm(lmf), fpm(st.wb), fpm(bb), count.data_ptr<int64_t>(), prologue_words(dev).data_ptr()));
            mm = moment_mask;
            if (n_known < 0) {
                host_n = at::empty({1}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
                host_n.copy_(count, /*non_blocking=*/true);
            }
        } else {
            st.wb = at::stack({loc[2].view({D}), loc[4].view({D}), loc[6].view({D})});
            bb = at::cat({loc[3], loc[5], loc[7]});
            // ---- layout, part 1: the cell count leaves for the host now and is waited for after the backbone is queued -- unless the
            // caller already knows it (flags >> 16 = count + 1; a captured step: nothing inside may wait for the device)
            mm = moment_mask.scalar_type() == at::kBool ? moment_mask : moment_mask.ne(0);
            if (n_known < 0) {
                host_n = at::empty({1}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
                host_n.copy_(mm.sum().reshape({1}), /*non_blocking=*/true);
            }
            st.len32 = qm.sum(1).to(at::kInt);
            st.vmaskf = cont(fl(video_mask.reshape({Bq * Tn})));
            qmf = cont(fl(qm)); lmf = cont(fl(length_mask));
        }
        hipEvent_t count_ready = next_event();
        TORCH_CHECK(hipEventRecord(count_ready, curs.stream()) == hipSuccess, "hipEventRecord failed");

        // ---- the video encoder's projection (models.py:25-36) on the second stream, beside the query encoder: only its product with the
        // sentence feature (models.py:81-83) waits for the LSTM layers (the fused call sat behind them: ~90 us of the step's opening chain).
        // Queued behind the parameter products: ahead of them it runs beside the LSTM operand packing and the first recurrence and
        // stretches both (pack 20 -> 84 us, recurrence 87 -> 130 us: the opening chain 45 us longer, tools/gantt.sh).
        st.vx = cont(video_features);
        st.fv = at::empty({Bq, T, (int64_t)D}, opt);
        Tensor f = at::empty({Bq, T, (int64_t)D}, opt);
        hipEvent_t projection_ready = nullptr;
        if (prep != curs) {
            await(prep, count_ready);                                              // (vmaskf)
            StreamScope sc(prep);
            SMIN_CK(smin_video_encoder_fwd(cur(), fp(st.vx), fp(all[P_VE_W]), fp(all[P_VE_B]), fp(all[P_PE]), fp(st.vmaskf), nullptr, B, Ti, i32(st.vx.size(2)), D, fpm(st.fv),
                                           nullptr));
            projection_ready = mark(prep);
        }

        // ---- backbone (models.py:38-83): BiLSTM x 2, sentence feature, fused video encoder
        Tensor x = cont(query_features);
        // both layers' operand layouts in one launch, ahead of the first recurrence (a launch per layer sat between the two)
        Tensor lstm_bias[2], lstm_W4[2];
        {
            const float* raw[16]; int ins[2]; float *wih[2], *bs[2], *whh[2], *w4[2];
            for (int layer = 0; layer < 2; ++layer) {
                LstmState& ls = st.lstm[layer];
                const int64_t In = layer == 0 ? x.size(2) : 2 * H;
                ls.Wih = at::empty({8 * H, In}, opt);                              // [w_ih; w_ih_reverse]
                lstm_bias[layer] = at::empty({8 * H}, opt);                        // b_ih + b_hh per direction
                ls.Whh = at::empty({2, 4 * H, H}, opt);
                lstm_W4[layer] = at::empty({2, H, H, 4}, opt);                     // [d, k, u, gate]
                for (int q = 0; q < 8; ++q) raw[8 * layer + q] = fp(all[P_LSTM + 8 * layer + q]);
                ins[layer] = i32(In); wih[layer] = fpm(ls.Wih); bs[layer] = fpm(lstm_bias[layer]); whh[layer] = fpm(ls.Whh); w4[layer] = fpm(lstm_W4[layer]);
            }
            SMIN_CK(smin_lstm_pack_layers(cur(), 2, raw, ins, i32(H), wih, bs, whh, w4));
        }
        for (int layer = 0; layer < 2; ++layer) {
            LstmState& ls = st.lstm[layer];
            const int In = i32(x.size(2)), Hh = i32(H);
            ls.x = x;
            const Tensor &bias = lstm_bias[layer], &W4 = lstm_W4[layer];
            ls.G = at::empty({Bq, Nq_in, 2, 4 * H}, opt); ls.Hout = at::empty({Bq, Nq_in, 2 * H}, opt); ls.Cs = at::empty({Bq, Nq_in, 2, H}, opt);
            SMIN_CK(smin_bilstm_layer_fwd(cur(), fp(x), fp(ls.Wih), fp(bias), fp(W4), ip(st.len32), B, i32(Nq_in), In, Hh, fpm(ls.G), fpm(ls.Hout), fpm(ls.Cs)));
            x = ls.Hout;
        }
        Tensor fw = x;
        if (Nq_in < maxq) fw = at::constant_pad_nd(fw, {0, 0, 0, maxq - Nq_in}, 0);
        fw = fw.contiguous();
        Tensor fs = at::empty({Bq, 2 * H}, opt);                                    // [h_fwd at the last word | h_bwd at the first word]
        SMIN_CK(smin_sentence_feature_fwd(cur(), fp(fw), ip(st.len32), B, i32(fw.size(1)), i32(H), fpm(fs)));
        if (projection_ready) {
            await(curs, projection_ready);
            SMIN_CK(smin_video_encoder_gate(cur(), fp(st.fv), fp(fs), B, Ti, D, fpm(f)));
        } else {
            SMIN_CK(smin_video_encoder_fwd(cur(), fp(st.vx), fp(all[P_VE_W]), fp(all[P_VE_B]), fp(all[P_PE]), fp(st.vmaskf), fp(fs), B, Ti, i32(st.vx.size(2)), D, fpm(st.fv), fpm(f)));
        }

        // ---- layout, part 2
        if (n_known < 0) TORCH_CHECK(hipEventSynchronize(count_ready) == hipSuccess, "hipEventSynchronize failed");
        const int64_t N = n_known < 0 ? host_n.const_data_ptr<int64_t>()[0] : n_known;
        const int n = i32(N);
        Tensor cells, row_ptr, cellmap;
        Tensor mask8 = mm.contiguous().view(at::kByte);
        hipEvent_t layout_ready;
        if (prep != curs) await(prep, count_ready);                               // (the mask; the second stream is idle while the LSTM runs)
        {
            StreamScope sc(prep);
            auto io = at::TensorOptions().dtype(at::kInt).device(dev);
            cells = at::empty({N, 4}, io); row_ptr = at::empty({Bq * L + 1}, io); cellmap = at::empty({Bq, L, L}, io);
            if (n_known < 0)
                SMIN_CK(smin_build_cells(cur(), static_cast<const uint8_t*>(mask8.const_data_ptr()), B, Li, 0, cells.data_ptr<int32_t>(), row_ptr.data_ptr<int32_t>(),
                                         cellmap.data_ptr<int32_t>()));
            else
                SMIN_CK(smin_build_cells_n(cur(), static_cast<const uint8_t*>(mask8.const_data_ptr()), B, Li, 0, n, cells.data_ptr<int32_t>(), row_ptr.data_ptr<int32_t>(),
                                           cellmap.data_ptr<int32_t>(), layout_status(dev).data_ptr<int32_t>()));
            layout_ready = mark(prep);
        }
        st.f = f; st.fw = fw; st.fs = fs; st.qmf = qmf; st.lmf = lmf; st.cells = cells; st.row_ptr = row_ptr; st.cellmap = cellmap;

        // ---- word-side operands on the second stream, behind the backbone; the main stream needs them at the first attention only
        hipEvent_t words_ready;
        wait_stream(prep, curs);
        {
            StreamScope sc(prep);
            std::vector<const float*> pp;
            for (int64_t k = 0; k < nl; ++k)
                for (int which : {L_WH_W, L_WH_B, L_SH_W, L_SH_B, L_AK_W, L_AK_B, L_AQ_W, L_AQ_B}) pp.push_back(fp(lp(k, which)));
            st.what = at::empty({nl, B, Nq, dl}, opt); st.kb = at::empty({nl, B, Nq, dl}, opt); st.Mq = at::empty({nl, B, Nq, dl}, opt);
            st.shat = at::empty({nl, B, dl}, opt); st.uq = at::empty({nl, B, Nq}, opt);
            SMIN_CK(smin_word_prep_fwd(cur(), fp(fw), fp(fs), fp(qmf), pp.data(), i32(nl), B, Nq, D, dl, fpm(st.what), fpm(st.shat), fpm(st.kb), fpm(st.Mq), fpm(st.uq)));
            words_ready = mark(prep);
        }
        if (prep != curs) { await(curs, products_ready); await(curs, layout_ready); }
        // Tensors cross streams here without recordStream bookkeeping (an event record and queries per tensor and step: ~0.3 ms of
        // host time).  What makes that safe: (1) every stretch of work on another stream starts with a wait for the main stream and
        // the main stream waits for every other stream before forward / backward return; (2) no tensor that another stream has
        // touched is released before that final wait (saved for backward, returned, or held in a function-scope list).  A block
        // therefore returns to its stream's pool only after all streams have met, and its next user is ordered behind that.

        // ---- proposal map (f_m, f_b) and every layer's clip-window term of chat
        Tensor fm = at::empty({N, D}, opt), fb = at::empty({B, L, D}, opt);
        {
            auto ws = scratch((size_t)8 * B * (T + 1) * std::max<int64_t>(D, nl * dl), dev);
            SMIN_CK(smin_proposal_map_fwd(cur(), fp(f), ip(cells), n, B, Ti, Li, Ci, D, nullptr, fpm(fm), fpm(fb), ws.p, ws.n));
        }
        Tensor pgs = at::empty({nl, N * C, dl}, opt);
        {
            Tensor g_all = at::empty({(int64_t)B * T, nl * dl}, opt);
            const float* xs[1] = {fp(f)};
            SMIN_CK(smin_linear_rows_fwd(cur(), xs, 1, fp(st.Wch_all), nullptr, nullptr, nullptr, 1, i32(B * T), i32(nl * dl), D, fpm(g_all)));
            auto ws = scratch((size_t)8 * B * (T + 1) * std::max<int64_t>(D, nl * dl), dev);
            SMIN_CK(smin_clip_window_means_fwd(cur(), fp(g_all), fp(st.layer[0].consts), dl, ip(cells), n, B, Ti, Li, Ci, dl, i32(nl), fpm(pgs), ws.p, ws.n));
        }

        Tensor cumean = fm, Hs;                                                    // mean_c f_c of the proposal map is f_m
        // Tensors that only ever feed contractions (the pair product, the attention outputs cc_k) are stored as bf16 when the contractions
        // round their operands to bf16 anyway (smin_set_gemm_mode(2)): no bit of the step changes, half the bytes
        const bool bf16_operands = (flags & F_BF16_OPERANDS) && smin_get_gemm_mode() == 2;
        const bool cc_bf16 = bf16_operands && dl % 8 == 0;
        for (int64_t k = 0; k < nl; ++k) {
            LayerState& ls = st.layer[k];
            const bool lastl = k == nl - 1;
            ls.fm = fm; ls.fb = fb;
            ls.hbar = at::empty_like(fm);
            ls.Hs = Hs;                                                            // sum of the earlier layers' gated features (undefined for k = 0)
            if (k > 0 && !lastl && N > 0) {                                        // the next layers' running sum comes out of the same pass
                Tensor Hs_next = at::empty_like(fm);
                SMIN_CK(smin_gate_fwd_sum(cur(), fp(fm), fp(fs), ip(cells), n, D, fpm(ls.hbar), fp(Hs), fpm(Hs_next)));
                Hs = Hs_next;
            } else {
                SMIN_CK(smin_gate_fwd(cur(), fp(fm), fp(fs), ip(cells), n, D, fpm(ls.hbar)));
                if (!lastl) Hs = Hs.defined() ? Hs + ls.hbar : ls.hbar;
            }
            // boundary unit on the second stream beside the content stream; joins before the moment unit
            wait_stream(side, curs);
            {
                StreamScope sc(side);
                ls.bu = at::empty_like(fb); ls.Qb = at::empty_like(fb); ls.baq = at::empty_like(fb); ls.bqv = at::empty_like(fb); ls.Kb = at::empty_like(fw);
                ls.P = at::empty({B, L, Nq}, opt); ls.A = at::empty({B, L, L}, opt);
                SMIN_CK(smin_boundary_unit_fwd(cur(), fp(fb), fp(fw), fp(fs), fp(ls.hbar), ip(cells), ip(row_ptr), n, B, Li, Nq, D, fp(lp(k, L_BQ_W)), fp(lp(k, L_BQ_B)),
                                               fp(lp(k, L_BK_W)), fp(lp(k, L_BK_B)), fp(qmf), fp(lmf), fpm(ls.bu), fpm(ls.Qb), fpm(ls.Kb), fpm(ls.P), fpm(ls.baq),
                                               fpm(ls.bqv), fpm(ls.A)));
            }
            // chat_k = clip-window term + [cc_0 | ..] Pcat^T + const_k + (Hs Wch^T per cell)
            Tensor chat = pgs[k];
            const Tensor& Hs_k = ls.Hs;
            for (int64_t part = 0, lo = 0; lo < k; ++part, lo += 4) {
                Tensor hp;
                if (lo == 0) {
                    hp = at::empty({N, dl}, opt);
                    const float* xs[1] = {fp(Hs_k)};
                    SMIN_CK(smin_linear_rows_fwd(cur(), xs, 1, fp(lp(k, L_CH_W)), nullptr, nullptr, nullptr, 1, n, dl, D, fpm(hp)));
                }
                const int nseg = i32(std::min<int64_t>(4, k - lo));
                Tensor y = at::empty({N * C, dl}, opt);
                if (cc_bf16) {
                    const uint16_t* xh[4];
                    for (int sgm = 0; sgm < nseg; ++sgm) xh[sgm] = hp16(st.layer[lo + sgm].cc);
                    SMIN_CK(smin_linear_rows_fwd_xh(cur(), xh, nseg, fp(ls.Pcat[part]), lo == 0 ? fp(ls.consts) : nullptr, fp(chat), fp(hp), Ci, i32(N * C), dl, dl, fpm(y)));
                } else {
                    const float* xs[4];
                    for (int sgm = 0; sgm < nseg; ++sgm) xs[sgm] = fp(st.layer[lo + sgm].cc);
                    SMIN_CK(smin_linear_rows_fwd(cur(), xs, nseg, fp(ls.Pcat[part]), lo == 0 ? fp(ls.consts) : nullptr, fp(chat), fp(hp), Ci, i32(N * C), dl, dl, fpm(y)));
                }
                chat = y;
            }
            ls.chat = chat;
            if (k == 0 && prep != curs) await(curs, words_ready);
            ls.cc = at::empty({lastl ? 0 : N * C, dl}, cc_bf16 ? opt.dtype(at::kBFloat16) : opt); ls.ccmean = at::empty({N, dl}, opt);
            if (cc_bf16 && !lastl)
                SMIN_CK(smin_content_attn_fwd_cch(cur(), fp(chat), ip(cells), ip(row_ptr), n, B, Li, Ci, dl, Nq, fp(st.Mq[k]), fp(st.uq[k]), fp(st.what[k]), fp(st.shat[k]),
                                                  fp(qmf), reinterpret_cast<uint16_t*>(ls.cc.data_ptr()), fpm(ls.ccmean)));
            else
                SMIN_CK(smin_content_attn_fwd(cur(), fp(chat), ip(cells), ip(row_ptr), n, B, Li, Ci, dl, Nq, fp(st.Mq[k]), fp(st.uq[k]), fp(st.what[k]), fp(st.shat[k]), fp(qmf),
                                              lastl ? nullptr : fpm(ls.cc), fpm(ls.ccmean)));
            ls.cum = at::empty({N, D}, opt);
            {
                const float* xs[1] = {fp(ls.ccmean)};
                SMIN_CK(smin_linear_rows_fwd(cur(), xs, 1, fp(lp(k, L_C_W)), fp(lp(k, L_C_B)), fp(cumean), fp(ls.hbar), 1, n, D, dl, fpm(ls.cum)));
            }
            wait_stream(curs, side);
            // f_b[i] * f_b[j], kept for the weight gradient (bf16 under bf16_operands: the layer's largest saved tensor)
            const bool x1h = bf16_operands;
            ls.x1 = x1h ? at::empty({N, D}, opt.dtype(at::kBFloat16)) : at::empty_like(fm);
            Tensor mu = at::empty_like(fm);
            if (x1h) {
                uint16_t* xh = reinterpret_cast<uint16_t*>(ls.x1.data_ptr());
                SMIN_CK(smin_pair_product_bf16(cur(), fp(ls.bu), ip(cells), n, Li, D, xh));
                SMIN_CK(smin_moment_unit_fwd_x1h(cur(), fp(ls.cum), fp(fm), fp(ls.bu), ip(cells), n, B, Li, D, fp(ls.Wcat), fp(bcat[k]), fpm(mu), xh));
            } else {
                SMIN_CK(smin_pair_product(cur(), fp(ls.bu), ip(cells), n, Li, D, fpm(ls.x1)));
                SMIN_CK(smin_moment_unit_fwd(cur(), fp(ls.cum), fp(fm), fp(ls.bu), ip(cells), n, B, Li, D, fp(ls.Wcat), fp(bcat[k]), fpm(mu), fp(ls.x1)));
            }
            fm = mu; cumean = ls.cum; fb = ls.bu;
        }
        // Localization (models.py:335-344)
        Tensor pm = at::empty({B, L, L}, opt), psea = at::empty({3, B, L}, opt);
        SMIN_CK(smin_score_map_fwd(cur(), fp(fm), fp(fb), ip(cells), n, B, Li, D, fp(loc[0]), fp(loc[1]), fp(st.wb), fp(bb), fp(lmf), fpm(pm), fpm(psea)));
        st.pm = pm; st.psea = psea; st.fm_out = fm;

        variable_list flat;
        visit_state(st, [&](Tensor& t) { flat.push_back(t); });
        for (auto& p : all) flat.push_back(p);
        ctx->save_for_backward(flat);
        ctx->saved_data["d"] = std::vector<int64_t>{N, T, L, C, nl, flags, H, Nq_in, prep_kernel ? 1 : 0};
        // ps / pe / pa leave as three outputs of the node (rows of one buffer), not as selections of one output: the selections'
        // backward nodes cost three zero fills, three copies and two adds between the loss and this node's backward
        return {pm, psea[0], psea[1], psea[2]};
    }

    static variable_list backward(AutogradContext* ctx, variable_list g)
    {
        auto d = ctx->saved_data["d"].toIntVector();
        const int64_t N = d[0], T = d[1], L = d[2], C = d[3], nl = d[4], flags = d[5], H = d[6], Nq_in = d[7];
        const bool prep_kernel = d[8] != 0;
        auto sv = ctx->get_saved_variables();
        CoreState st;
        size_state(st, nl);
        size_t cursor = 0;
        visit_state(st, [&](Tensor& t) { t = sv[cursor++]; });
        std::vector<Tensor> all(sv.begin() + cursor, sv.end());
        std::vector<Tensor> prm(all.begin() + P_LAYER0, all.end());
        auto lp = [&](int64_t k, int which) -> const Tensor& { return prm[k * L_COUNT + which]; };
        const Tensor* loc = &prm[nl * L_COUNT];
        const Tensor &f = st.f, &fw = st.fw, &fs = st.fs, &qmf = st.qmf, &lmf = st.lmf, &cells = st.cells, &row_ptr = st.row_ptr, &cellmap = st.cellmap;
        const at::Device dev = f.device();
        const auto opt = f.options();
        const int B = i32(f.size(0)), D = i32(f.size(2)), Nq = i32(fw.size(1)), dl = i32(prm[L_CH_W].size(0)), n = i32(N), Li = i32(L), Ci = i32(C), Ti = i32(T);
        HStream curs = c10::hip::getCurrentHIPStream(dev.index());
        HStream side = (flags & F_OVERLAP_BOUNDARY) ? side_stream(dev.index()) : curs;
        // weight-gradient contractions: nothing waits for them before the end of the step, so they go to a low-priority stream
        // of their own and fill the chip beside the bandwidth-bound kernels of the main chain
        HStream wstr = (flags & F_ASYNC_WEIGHTS) ? weight_stream(dev.index()) : curs;
        std::vector<Tensor> keep;                                                  // main-stream tensors read on wstr: alive until the streams join
        GradSync sync;
        sync.on = (flags & F_GRAD_SYNC) != 0;
        TORCH_CHECK(!sync.on || prep_kernel, "smin_forward: the in-node gradient exchange needs the parameter-product kernel (D % 32 == 0, D <= 1056, dl % 32 == 0, <= 8 layers)");
        TORCH_CHECK(!sync.on || !grad_sync_config().group.empty(), "smin_forward: grad_sync requested but no process group was set (smin_hip::set_grad_sync)");
        std::vector<Tensor> dprm(prm.size());
        auto dlp = [&](int64_t k, int which) -> Tensor& { return dprm[k * L_COUNT + which]; };
        auto acc = [](Tensor& into, const Tensor& t) { if (into.defined()) into.add_(t); else into = t; };

        // ---- W^T of every contraction weight, one launch
        enum { TR_CH = 0, TR_C, TR_CAT, TR_BQ, TR_BK, TR_PER_LAYER };
        std::vector<Tensor> tr_in;
        for (int64_t k = 0; k < nl; ++k) {
            tr_in.push_back(lp(k, L_CH_W)); tr_in.push_back(lp(k, L_C_W)); tr_in.push_back(st.layer[k].Wcat); tr_in.push_back(lp(k, L_BQ_W)); tr_in.push_back(lp(k, L_BK_W));
        }
        const size_t tr_pcat0 = tr_in.size();
        for (int64_t k = 0; k < nl; ++k) for (auto& p : st.layer[k].Pcat) tr_in.push_back(p);
        tr_in.push_back(st.Wch_all);
        tr_in.push_back(st.lstm[0].Wih); tr_in.push_back(st.lstm[1].Wih);
        // (on the boundary stream, as the boundary heads' backward below: beside the score map's backward on the main stream, which
        //  otherwise opens the backward pass with four short launches in a row in front of the first contraction)
        HStream early = (side != curs && (flags & F_OVERLAP_PREP)) ? side : curs;
        auto mark0 = [](HStream on) { hipEvent_t e = next_event(); TORCH_CHECK(hipEventRecord(e, on.stream()) == hipSuccess, "hipEventRecord failed"); return e; };
        wait_stream(early, curs);
        std::vector<Tensor> tr;
        {
            StreamScope sc(early);
            tr = transpose_all(tr_in);
        }
        auto trk = [&](int64_t k, int which) -> const Tensor& { return tr[k * TR_PER_LAYER + which]; };
        std::vector<std::vector<Tensor>> PcatT(nl);
        { size_t i = tr_pcat0; for (int64_t k = 0; k < nl; ++k) for (size_t p = 0; p < st.layer[k].Pcat.size(); ++p) PcatT[k].push_back(tr[i++]); }
        const Tensor &Wch_allT = tr[tr.size() - 3], *WihT = &tr[tr.size() - 2];

        // ---- Localization
        const Tensor& bu_last = st.layer[nl - 1].bu;
        Tensor dpm = g[0].defined() ? cont(g[0]) : at::zeros_like(st.pm), dpsea;
        {
            // the three score gradients as one (3, B, L) buffer: the library's loss hands over rows of one buffer (used as it is);
            // anything else is gathered
            const int64_t BL = (int64_t)B * L;
            bool rows = g[1].defined() && g[2].defined() && g[3].defined();
            for (int h = 1; rows && h <= 3; ++h) rows = g[h].scalar_type() == at::kFloat && g[h].is_contiguous() && g[h].numel() == BL;
            if (rows && fp(g[2]) == fp(g[1]) + BL && fp(g[3]) == fp(g[1]) + 2 * BL) dpsea = g[1];
            else {
                std::vector<Tensor> parts;
                for (int h = 1; h <= 3; ++h) parts.push_back(g[h].defined() ? fl(g[h]).reshape({B, L}) : at::zeros({B, L}, opt));
                dpsea = at::stack(parts);
            }
        }
        Tensor dfm = at::empty({N, D}, opt), dfb_next = at::empty({B, L, D}, opt);
        std::vector<Tensor> loc_bufs;
        {
            Tensor dwm = at::empty({D}, opt), dbm = at::empty({1}, opt), dwb = at::empty({3, D}, opt), dbb = at::empty({3}, opt);
            if (early != curs) {
                wait_stream(early, curs);                                          // (behind the allocations above: see "the backward's tail", lesson 2)
                {
                    StreamScope sc(early);
                    auto ws = scratch(smin_workspace_bytes(n, B, 4, D, 4, 1), dev);
                    SMIN_CK(smin_score_map_bwd(cur(), nullptr, fp(dpsea), fp(st.pm), fp(st.psea), fp(st.fm_out), fp(bu_last), ip(cells), n, B, Li, D, fp(loc[0]), fp(st.wb),
                                               fp(lmf), nullptr, fpm(dfb_next), nullptr, nullptr, fpm(dwb), fpm(dbb), ws.p, ws.n));
                }
                hipEvent_t early_done = mark0(early);
                auto ws = scratch(smin_workspace_bytes(n, B, 4, D, 4, 1), dev);
                SMIN_CK(smin_score_map_bwd(cur(), fp(dpm), nullptr, fp(st.pm), fp(st.psea), fp(st.fm_out), fp(bu_last), ip(cells), n, B, Li, D, fp(loc[0]), fp(st.wb), fp(lmf),
                                           fpm(dfm), nullptr, fpm(dwm), fpm(dbm), nullptr, nullptr, ws.p, ws.n));
                TORCH_CHECK(hipStreamWaitEvent(curs.stream(), early_done, 0) == hipSuccess, "hipStreamWaitEvent failed");    // W^T and dfb_next
            } else {
                auto ws = scratch(smin_workspace_bytes(n, B, 4, D, 4, 1), dev);
                SMIN_CK(smin_score_map_bwd(cur(), fp(dpm), fp(dpsea), fp(st.pm), fp(st.psea), fp(st.fm_out), fp(bu_last), ip(cells), n, B, Li, D, fp(loc[0]), fp(st.wb), fp(lmf),
                                           fpm(dfm), fpm(dfb_next), fpm(dwm), fpm(dbm), fpm(dwb), fpm(dbb), ws.p, ws.n));
            }
            Tensor* dloc = &dprm[nl * L_COUNT];
            loc_bufs = {dwm, dbm, dwb, dbb};
            dloc[0] = dwm.view_as(loc[0]); dloc[1] = dbm.view_as(loc[1]);
            for (int h = 0; h < 3; ++h) { dloc[2 + 2 * h] = dwb[h].view_as(loc[2 + 2 * h]); dloc[3 + 2 * h] = dbb.slice(0, h, h + 1).view_as(loc[3 + 2 * h]); }
        }

        std::vector<Tensor> dcc(nl), dHs(nl), dchat(nl), dconsts(nl), dfs_parts, dfw_parts;
        // prep_kernel: the gradients that meet in the parameter products are gathered as the kernel of csrc/param_prep.hip wants them
        Tensor dconsts_all, dWcat_all, dbcat_all, dWch_all;
        std::vector<Tensor> base_ch(nl), base_c(nl), base_bc(nl);
        if (prep_kernel) {
            dconsts_all = at::empty({nl, dl}, opt); dWcat_all = at::empty({nl, D, 2 * D}, opt); dbcat_all = at::empty({nl, D}, opt);
            for (int64_t k = 0; k < nl; ++k) dconsts[k] = dconsts_all[k];
        }
        std::vector<std::vector<Tensor>> dPcat(nl);
        Tensor dwhat = at::empty_like(st.what), dshat = at::empty_like(st.shat), dMq = at::empty_like(st.Mq), duq = at::empty_like(st.uq);
        if (N == 0) { dwhat.zero_(); dshat.zero_(); dMq.zero_(); duq.zero_(); }
        Tensor dcum_next;                                                          // gradient of cum_k from layer k+1's clip-mean chain
        auto mark_on = [](HStream on) { hipEvent_t e = next_event(); TORCH_CHECK(hipEventRecord(e, on.stream()) == hipSuccess, "hipEventRecord failed"); return e; };
        hipEvent_t attn0_done = nullptr;
        hipEvent_t boundary_done = nullptr;                                        // the previous iteration's boundary-unit backward (side stream)
        static const bool defer_dw0 = std::getenv("SMIN_DEFER_DW0") && std::atoi(std::getenv("SMIN_DEFER_DW0")) != 0;
        Tensor deferred_dfm;
        std::function<void(const Tensor&)> deferred_weights;
        for (int64_t k = nl - 1; k >= 0; --k) {
            LayerState& ls = st.layer[k];
            // moment unit: dmu -> d cum (its chain gradient folded in), d bu, weight gradients; the residual gradient is dmu itself
            Tensor dcum = at::empty({N, D}, opt), dfb_mu = at::empty({B, L, D}, opt);
            keep.push_back(dfm); keep.push_back(dcum);
            // (measured: the weight half queued ahead of the input half 21.39 -> 21.15 ms/step, behind it 21.7 -> 21.6)
            // SMIN_DEFER_DW0=1 (experiment, off): layer 0's (the last one of the loop) held back until the proposal map's gradient is
            // queued, so that it does not stretch the HBM-bound closing kernels.  Measured WORSE (17.8 vs 17.6 ms): it then runs beside
            // the cluster LSTM, whose workgroups need a CU's whole LDS and wait for CUs this contraction has drained (177 -> 1225 us).
            auto moment_weights = [&, k](const Tensor& dfm_in) {
                LayerState& lsk = st.layer[k];
                wait_stream(wstr, curs);
                StreamScope sc(wstr);
                Tensor dWcat = prep_kernel ? dWcat_all[k] : at::empty_like(lsk.Wcat), dbcat = prep_kernel ? dbcat_all[k] : at::empty({D}, opt);
                auto ws = scratch(smin_workspace_bytes(n, B, 4, D, 4, 1), dev);
                if (lsk.x1.scalar_type() == at::kBFloat16)
                    SMIN_CK(smin_moment_unit_bwd_x1h(cur(), fp(dfm_in), fp(lsk.cum), fp(lsk.bu), ip(cells), ip(row_ptr), ip(cellmap), n, B, Li, D, fp(trk(k, TR_CAT)), nullptr, nullptr,
                                                     fpm(dWcat), fpm(dbcat), ws.p, ws.n, 1, nullptr, reinterpret_cast<const uint16_t*>(lsk.x1.const_data_ptr()), nullptr));
                else
                    SMIN_CK(smin_moment_unit_bwd(cur(), fp(dfm_in), fp(lsk.cum), fp(lsk.bu), ip(cells), ip(row_ptr), ip(cellmap), n, B, Li, D, fp(trk(k, TR_CAT)), nullptr, nullptr,
                                                 fpm(dWcat), fpm(dbcat), ws.p, ws.n, 1, nullptr, fp(lsk.x1), nullptr));
                if (!prep_kernel) {
                    dlp(k, L_FB_W) = dWcat.slice(1, 0, D).contiguous().view_as(lp(k, L_FB_W)); dlp(k, L_FC_W) = dWcat.slice(1, D).contiguous().view_as(lp(k, L_FC_W));
                    dlp(k, L_FB_B) = dbcat; dlp(k, L_FC_B) = dbcat;
                }
                sync.reduce({dWcat, dbcat}, wstr);                                   // inputs of the parameter-product kernel
            };
            if (k == 0 && defer_dw0 && wstr != curs) { deferred_dfm = dfm; deferred_weights = moment_weights; }
            else moment_weights(dfm);
            // the previous layer's boundary-unit backward (second stream) is awaited HERE, where its dfb is first read -- not in front of
            // that layer's gate backward, which reads nothing of it any more (it forms the unit's dhbar itself): the main stream sat
            // ~0.6 ms behind the unit's weight contractions at the end of layer 0 (tools/gantt.sh)
            if (boundary_done) { TORCH_CHECK(hipStreamWaitEvent(curs.stream(), boundary_done, 0) == hipSuccess, "hipStreamWaitEvent failed"); boundary_done = nullptr; }
            {
                auto ws = scratch(smin_workspace_bytes(n, B, 4, D, 4, 1), dev);
                SMIN_CK(smin_moment_unit_bwd(cur(), fp(dfm), fp(ls.cum), fp(ls.bu), ip(cells), ip(row_ptr), ip(cellmap), n, B, Li, D, fp(trk(k, TR_CAT)), fpm(dcum), fpm(dfb_mu),
                                             nullptr, nullptr, ws.p, ws.n, 1, fp(dcum_next), nullptr, fp(dfb_next)));   // (the pair product feeds the weight half only)
            }                                                                      // dfb_mu = this unit's gradient of bu + the later consumer's (dfb_next): dbu
            // boundary unit on the second stream
            // (the boundary unit's gradient of hbar, A[b,i,j] * dbu[b,i,:], is formed by the gate backward itself from A and dbu: no [N][D]
            //  tensor written here and read back there)
            Tensor dfb_k, dbu;
            wait_stream(side, curs);
            {
                StreamScope sc(side);
                dbu = dfb_mu;
                dfb_k = at::empty({B, L, D}, opt);
                Tensor dfw = at::empty_like(fw), dfs = at::empty_like(fs);
                Tensor dWq = at::empty({D, D}, opt), dbq = at::empty({D}, opt), dWk = at::empty({D, D}, opt), dbk = at::empty({D}, opt);
                const size_t nbytes = 4 * ((size_t)2 * B * L * L + (size_t)3 * B * L * D + (size_t)B * L * Nq + (size_t)B * Nq * D + (size_t)2 * 64 * ((size_t)D * D + D)) + 4096;
                auto ws = scratch(nbytes, dev);
                SMIN_CK(smin_boundary_unit_bwd(cur(), fp(dbu), fp(ls.fb), fp(fw), fp(fs), fp(ls.hbar), ip(cells), ip(row_ptr), n, B, Li, Nq, D, fp(trk(k, TR_BQ)), fp(trk(k, TR_BK)),
                                               fp(qmf), fp(lmf), fp(ls.Qb), fp(ls.Kb), fp(ls.P), fp(ls.baq), fp(ls.bqv), fp(ls.A), fpm(dfb_k), fpm(dfw), fpm(dfs), nullptr,
                                               fpm(dWq), fpm(dbq), fpm(dWk), fpm(dbk), ws.p, ws.n));
                dlp(k, L_BQ_W) = dWq; dlp(k, L_BQ_B) = dbq; dlp(k, L_BK_W) = dWk; dlp(k, L_BK_B) = dbk;
                sync.reduce({dWq, dbq, dWk, dbk}, side);
                dfs_parts.push_back(dfs); dfw_parts.push_back(dfw);
                keep.push_back(dfb_next); keep.push_back(dfb_mu); keep.push_back(dbu);
            }
            if (side != curs) boundary_done = mark_on(side);
            // clip-mean update cum = ccmean Wc^T + b + cumean + hbar: d ccmean, weight gradients; d cumean = d hbar = dcum
            Tensor dccmean = at::empty({N, dl}, opt);
            {
                const float* xs[1] = {fp(ls.ccmean)}; float* dxs[1] = {fpm(dccmean)};
                auto ws = scratch(smin_linear_rows_bwd_workspace_bytes(n, D, dl), dev);
                SMIN_CK(smin_linear_rows_bwd(cur(), fp(dcum), xs, 1, fp(trk(k, TR_C)), n, D, dl, dxs, nullptr, nullptr, ws.p, ws.n));
            }
            // attention core
            dchat[k] = at::empty({N * C, dl}, opt);
            if (N > 0) {
                auto ws = scratch(smin_content_attn_bwd_workspace_bytes(n, B, Ci, dl), dev);
                SMIN_CK(smin_content_attn_bwd(cur(), fp(dcc[k]), fp(dccmean), fp(ls.chat), ip(cells), ip(row_ptr), n, B, Li, Ci, dl, Nq, fp(st.Mq[k]), fp(st.uq[k]), fp(st.what[k]),
                                              fp(st.shat[k]), fp(qmf), fpm(dchat[k]), fpm(dMq[k]), fpm(duq[k]), fpm(dwhat[k]), fpm(dshat[k]), ws.p, ws.n));
            }
            if (k == 0) attn0_done = mark_on(curs);                                // every dchat / word-side gradient is final from here on
            // chat_k's contraction over the earlier layers' attention outputs, and its per-cell gate term: input gradients
            Tensor dhp;
            for (int64_t part = 0, lo = 0; lo < k; ++part, lo += 4) {
                const int nseg = i32(std::min<int64_t>(4, k - lo));
                const float* xs[4]; float* dxs[4];
                // every earlier layer's attention output already has the gradient of the later layers (all of them, or none for the last
                // layer): accumulate in the epilogue instead of a full-size add per tensor afterwards
                const bool have = dcc[lo].defined();
                for (int sgm = 0; sgm < nseg; ++sgm) {
                    TORCH_CHECK(dcc[lo + sgm].defined() == have, "content stream: inconsistent gradient state of the attention outputs");
                    xs[sgm] = nullptr;                                             // (the input gradients never read the operands)
                    if (!have) dcc[lo + sgm] = at::empty({N * C, dl}, opt);
                    dxs[sgm] = fpm(dcc[lo + sgm]);
                }
                if (have) {
                    SMIN_CK(smin_linear_rows_dx_acc(cur(), fp(dchat[k]), nseg, fp(PcatT[k][part]), i32(N * C), dl, dl, dxs));
                } else {
                    auto ws = scratch(smin_linear_rows_bwd_workspace_bytes(i32(N * C), dl, nseg * dl), dev);
                    SMIN_CK(smin_linear_rows_bwd(cur(), fp(dchat[k]), xs, nseg, fp(PcatT[k][part]), i32(N * C), dl, dl, dxs, nullptr, nullptr, ws.p, ws.n));
                }
            }
            if (k > 0) {
                dhp = at::empty({N, dl}, opt);
                keep.push_back(dhp);
                SMIN_CK(smin_group_sum(cur(), fp(dchat[k]), n, Ci, dl, fpm(dhp)));
                dHs[k] = at::empty({N, D}, opt);
                const float* xs[1] = {fp(ls.Hs)}; float* dxs[1] = {fpm(dHs[k])};
                auto ws = scratch(smin_linear_rows_bwd_workspace_bytes(n, dl, D), dev);
                SMIN_CK(smin_linear_rows_bwd(cur(), fp(dhp), xs, 1, fp(trk(k, TR_CH)), n, dl, D, dxs, nullptr, nullptr, ws.p, ws.n));
            }
            // ... and this layer's weight gradients of the content stream
            wait_stream(wstr, curs);
            {
                StreamScope sc(wstr);
                {
                    Tensor dWc = at::empty_like(lp(k, L_C_W)), dbc = at::empty({D}, opt);
                    const float* xs[1] = {fp(ls.ccmean)};
                    auto ws = scratch(smin_linear_rows_bwd_workspace_bytes(n, D, dl), dev);
                    SMIN_CK(smin_linear_rows_bwd(cur(), fp(dcum), xs, 1, nullptr, n, D, dl, nullptr, fpm(dWc), fpm(dbc), ws.p, ws.n));
                    if (prep_kernel) { base_c[k] = dWc; base_bc[k] = dbc; }
                    else { acc(dlp(k, L_C_W), dWc); acc(dlp(k, L_C_B), dbc); }
                }
                for (int64_t part = 0, lo = 0; lo < k; ++part, lo += 4) {
                    const int nseg = i32(std::min<int64_t>(4, k - lo));
                    Tensor dP = at::empty_like(ls.Pcat[part]);
                    if (lo == 0 && !prep_kernel) dconsts[k] = at::empty({dl}, opt);
                    auto ws = scratch(smin_linear_rows_bwd_workspace_bytes(i32(N * C), dl, nseg * dl), dev);
                    if (st.layer[lo].cc.scalar_type() == at::kBFloat16) {
                        const uint16_t* xh[4];
                        for (int sgm = 0; sgm < nseg; ++sgm) xh[sgm] = hp16(st.layer[lo + sgm].cc);
                        SMIN_CK(smin_linear_rows_bwd_xh(cur(), fp(dchat[k]), xh, nseg, i32(N * C), dl, dl, fpm(dP), lo == 0 ? fpm(dconsts[k]) : nullptr, ws.p, ws.n));
                    } else {
                        const float* xs[4];
                        for (int sgm = 0; sgm < nseg; ++sgm) xs[sgm] = fp(st.layer[lo + sgm].cc);
                        SMIN_CK(smin_linear_rows_bwd(cur(), fp(dchat[k]), xs, nseg, nullptr, i32(N * C), dl, dl, nullptr, fpm(dP), lo == 0 ? fpm(dconsts[k]) : nullptr, ws.p, ws.n));
                    }
                    dPcat[k].push_back(dP);
                }
                if (k > 0) {
                    Tensor dWch = at::empty_like(lp(k, L_CH_W));
                    const float* xs[1] = {fp(ls.Hs)};
                    auto ws = scratch(smin_linear_rows_bwd_workspace_bytes(n, dl, D), dev);
                    SMIN_CK(smin_linear_rows_bwd(cur(), fp(dhp), xs, 1, nullptr, n, dl, D, nullptr, fpm(dWch), nullptr, ws.p, ws.n));
                    if (prep_kernel) base_ch[k] = dWch; else acc(dlp(k, L_CH_W), dWch);
                }
            }
            // gate: every consumer of hbar_k (clip-mean update, boundary unit, the later layers' running sums) and of f_m (residual; layer 0: the clip-mean chain)
            {
                std::vector<Tensor> later;
                for (int64_t kk = k + 1; kk < nl; ++kk) later.push_back(dHs[kk]);
                std::vector<const float*> dh{fp(dcum)}, dr{fp(dfm)};
                Tensor later_sum;
                if (later.size() <= 3) for (auto& t : later) dh.push_back(fp(t));
                else { later_sum = sum_list(later); dh.push_back(fp(later_sum)); }
                if (k == 0) dr.push_back(fp(dcum));
                Tensor dfm_k = at::empty({N, D}, opt), dfs = at::empty_like(fs);
                auto ws = scratch((size_t)4 * B * 512 * D + 4096, dev);
                SMIN_CK(smin_gate_bwd(cur(), dh.data(), i32(dh.size()), dr.data(), i32(dr.size()), fp(ls.fm), fp(fs), ip(row_ptr), n, B, Li, D, fpm(dfm_k), fpm(dfs), ws.p, ws.n,
                                      ip(cells), fp(ls.A), fp(dbu)));
                dfs_parts.push_back(dfs);
                dfm = dfm_k;
            }
            dcum_next = dcum; dfb_next = dfb_k;
        }

        // ---- the tail.  Three chains leave the last attention backward (layer 0's) and meet again in front of the LSTM layers:
        //   words  (own stream): the word-side operands' backward, 0.75 ms of small-grid kernels at ActivityNet size -> d f_w
        //   tail   (the boundary stream): the clip-window gradients of every layer's chat -> dg; later the parameter products
        //   main   : layer 0's gate backward (already queued above), the proposal map's gradient -> df, video encoder, LSTM layers
        // (before: the first two waited for the gate backward and the clip-window pass sat between it and the proposal map on the
        //  main stream -- 0.5 ms longer, tools/gantt.sh)
        // F_NO_TAIL_SPLIT / SMIN_TAIL_SPLIT=0: the round-2 placement (words on the boundary stream, clip-window gradients on the main
        // stream).  training.CapturedStep asks for it: a replayed graph pays for the extra streams (tacos.yml captured: 3.5 ms/step
        // once the process has used them, 2.5 without; eager 2.2) -- and never inside a stream capture.
        static const bool tail_split_env = !(std::getenv("SMIN_TAIL_SPLIT") && std::atoi(std::getenv("SMIN_TAIL_SPLIT")) == 0);
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        TORCH_CHECK(hipStreamIsCapturing(curs.stream(), &capture) == hipSuccess, "hipStreamIsCapturing failed");
        const bool tail_split = tail_split_env && !(flags & F_NO_TAIL_SPLIT) && capture == hipStreamCaptureStatusNone;
        HStream tail = (flags & F_OVERLAP_PREP) ? side_stream(dev.index()) : curs;
        HStream wordst = !(flags & F_OVERLAP_PREP) ? curs : tail_split ? side_stream(dev.index(), 1) : tail;
        HStream cw = tail_split ? tail : curs;                                      // stream of the clip-window gradients
        auto mark = mark_on;
        auto await = [](HStream waiter, hipEvent_t e) { TORCH_CHECK(hipStreamWaitEvent(waiter.stream(), e, 0) == hipSuccess, "hipStreamWaitEvent failed"); };
        if (!attn0_done) attn0_done = mark(curs);
        hipEvent_t words_done;
        if (!tail_split) wait_stream(wordst, curs);
        else if (wordst != curs) await(wordst, attn0_done);
        {
            StreamScope sc(wordst);
            std::vector<const float*> gp[4], pp;
            std::vector<float*> dp;
            for (int64_t k = 0; k < nl; ++k) {
                gp[0].push_back(fp(dwhat[k])); gp[1].push_back(fp(dshat[k])); gp[2].push_back(fp(dMq[k])); gp[3].push_back(fp(duq[k]));
                for (int which : {L_WH_W, L_WH_B, L_SH_W, L_SH_B, L_AK_W, L_AK_B, L_AQ_W, L_AQ_B}) {
                    pp.push_back(fp(lp(k, which)));
                    dlp(k, which) = at::empty_like(lp(k, which));
                    dp.push_back(fpm(dlp(k, which)));
                }
            }
            Tensor dfw = at::empty_like(fw), dfs = at::empty_like(fs);
            auto ws = scratch(smin_word_prep_bwd_workspace_bytes(i32(nl), B, Nq, D, dl), dev);
            SMIN_CK(smin_word_prep_bwd(cur(), gp[0].data(), gp[1].data(), gp[2].data(), gp[3].data(), fp(fw), fp(fs), fp(qmf), fp(st.what), fp(st.kb), pp.data(), i32(nl), B, Nq, D,
                                       dl, fpm(dfw), fpm(dfs), dp.data(), ws.p, ws.n));
            dfw_parts.push_back(dfw); dfs_parts.push_back(dfs);
            words_done = mark(wordst);
            if (sync.on) {
                std::vector<Tensor> ws_grads = loc_bufs;
                for (int64_t k = 0; k < nl; ++k)
                    for (int which : {L_WH_W, L_WH_B, L_SH_W, L_SH_B, L_AK_W, L_AK_B, L_AQ_W, L_AQ_B}) ws_grads.push_back(dlp(k, which));
                sync.reduce(ws_grads, wordst);
            }
        }

        // ---- clip-window terms of chat (tail stream), the proposal map (main stream), f
        Tensor df;
        hipEvent_t weights_done;
        bool tab_built = false;
        auto tab = clip_event_table(dev, Ti, Li, Ci, &tab_built);
        if (tab_built) wait_stream(cw, curs);                                    // first backward of this geometry only
        {
            std::vector<const float*> ptrs;
            for (int64_t k = 0; k < nl; ++k) ptrs.push_back(fp(dchat[k]));
            if (cw != curs) await(cw, attn0_done);
            hipEvent_t dg_done;
            Tensor dg;
            {
                StreamScope sc(cw);
                // allocated under THIS stream: a block from the main stream's pool may still be read by main-stream kernels queued behind
                // attn0_done (the allocator only orders reuse within the stream a block was allocated on)
                dg = at::empty({(int64_t)B * T, nl * dl}, opt);
                keep.push_back(dg);
                auto ws = scratch((size_t)4 * B * T * std::max<int64_t>(D, nl * dl), dev);
                SMIN_CK(smin_clip_window_means_bwd(cur(), ptrs.data(), ip(cells), ip(row_ptr), ip(cellmap), n, B, Ti, Li, Ci, dl, i32(nl), fpm(dg), ws.p, ws.n, ip(tab.first),
                                                   tab.second.data_ptr()));
                dg_done = mark(cw);
            }
            if (!tail_split) wait_stream(wstr, curs);
            {
                StreamScope sc(tail_split ? tail : wstr);
                // the weight gradient of the clip-window contraction stays on this stream (its only consumer is the parameter-product kernel
                // queued here below).  NOT on the weight stream: that stream would wait for this one and this one for it, and two forked
                // streams that wait for each other's events send hipStreamEndCapture into an endless recursion (captured step).
                if (!prep_kernel) dconsts[0] = at::empty({dl}, opt);
                const float* xs[1] = {fp(f)};
                dWch_all = at::empty_like(st.Wch_all);
                auto wsw = scratch(smin_linear_rows_bwd_workspace_bytes(i32(B * T), i32(nl * dl), D), dev);
                SMIN_CK(smin_linear_rows_bwd(cur(), fp(dg), xs, 1, nullptr, i32(B * T), i32(nl * dl), D, nullptr, fpm(dWch_all), nullptr, wsw.p, wsw.n));
                if (!prep_kernel) for (int64_t k = 0; k < nl; ++k) acc(dlp(k, L_CH_W), dWch_all.slice(0, k * dl, (k + 1) * dl));
                // layer 0's constant: its gradient is the column sum of dchat_0 (the later layers' ride on their weight-gradient passes)
                auto wsc = scratch(smin_col_sum_workspace_bytes(i32(N * C), dl), dev);
                SMIN_CK(smin_col_sum(cur(), fp(dchat[0]), i32(N * C), dl, fpm(dconsts[0]), wsc.p, wsc.n));
            }
            // df = gradient through the proposal map (f_m, f_b) + gradient through the clip-window terms, the second accumulated by its
            // contraction's epilogue
            df = at::empty({B, T, D}, opt);
            if (boundary_done) { TORCH_CHECK(hipStreamWaitEvent(curs.stream(), boundary_done, 0) == hipSuccess, "hipStreamWaitEvent failed"); boundary_done = nullptr; }   // layer 0's dfb
            auto ws3 = scratch((size_t)4 * B * T * std::max<int64_t>(D, nl * dl), dev);
            SMIN_CK(smin_proposal_map_bwd(cur(), nullptr, fp(dfm), fp(dfb_next), ip(cells), ip(row_ptr), ip(cellmap), n, B, Ti, Li, Ci, D, fpm(df), ws3.p, ws3.n, ip(tab.first),
                                          tab.second.data_ptr()));
            if (deferred_weights) { deferred_weights(deferred_dfm); deferred_weights = nullptr; }
            weights_done = mark(wstr);
            float* dxs[1] = {fpm(df)};
            if (cw != curs) await(curs, dg_done);
            SMIN_CK(smin_linear_rows_dx_acc(cur(), fp(dg), 1, fp(Wch_allT), i32(B * T), i32(nl * dl), D, dxs));
        }

        // ---- parameter products on the second stream: consts_k = b_ch_k + Wch_k bsum_k (bsum_k = sum_{l<k} b_c_l), Pcat_k = [Wch_k Wc_l]_l
        if (tail != wstr) await(tail, weights_done);
        if (sync.on) {
            std::vector<Tensor> ins{dconsts_all, dWch_all};
            for (int64_t k = 0; k < nl; ++k) {
                ins.push_back(base_ch[k]); ins.push_back(base_c[k]); ins.push_back(base_bc[k]);
                for (auto& t : dPcat[k]) ins.push_back(t);
            }
            sync.reduce(ins, tail);
            sync.join(tail);                                                        // every input of the parameter products (the layers' dWcat included) is averaged
        }
        if (prep_kernel) {
            StreamScope sc(tail);
            std::vector<const float*> pp, dpc(nl * 2, nullptr), bch(nl, nullptr), bc(nl), bbc(nl);
            std::vector<float*> gp;
            for (int64_t k = 0; k < nl; ++k) {
                for (int which : {L_CH_W, L_CH_B, L_C_W, L_C_B, L_FB_W, L_FB_B, L_FC_W, L_FC_B}) {
                    pp.push_back(fp(lp(k, which)));
                    dlp(k, which) = at::empty_like(lp(k, which));
                    gp.push_back(fpm(dlp(k, which)));
                }
                for (size_t part = 0; part < dPcat[k].size(); ++part) dpc[k * 2 + part] = fp(dPcat[k][part]);
                bch[k] = fp(base_ch[k]); bc[k] = fp(base_c[k]); bbc[k] = fp(base_bc[k]);
            }
            SMIN_CK(smin_param_prep_bwd(cur(), pp.data(), i32(nl), D, dl, dpc.data(), fp(dconsts_all), fp(dWcat_all), fp(dbcat_all), fp(dWch_all), bch.data(), bc.data(),
                                        bbc.data(), gp.data()));
        } else {
            StreamScope sc(tail);
            Tensor bsum;
            std::vector<Tensor> bsums(nl);
            for (int64_t k = 0; k < nl; ++k) { bsums[k] = bsum; bsum = bsum.defined() ? bsum + lp(k, L_C_B) : lp(k, L_C_B); }
            Tensor dbsum_run;                                                       // sum_{k' > l} Wch_k'^T dconst_k'
            for (int64_t k = nl - 1; k >= 0; --k) {
                dlp(k, L_CH_B) = dconsts[k];
                if (dbsum_run.defined()) dlp(k, L_C_B).add_(dbsum_run);           // in place: the buffer was allocated on the main stream and must outlive this launch
                if (k > 0) {
                    dlp(k, L_CH_W).addr_(dconsts[k], bsums[k]);
                    Tensor dbs = at::mv(trk(k, TR_CH), dconsts[k]);
                    dbsum_run = dbsum_run.defined() ? dbsum_run + dbs : dbs;
                }
                for (size_t part = 0; part < dPcat[k].size(); ++part)
                    for (int64_t l = (int64_t)part * 4; l < std::min<int64_t>((int64_t)part * 4 + 4, k); ++l) {
                        Tensor dP = dPcat[k][part].slice(1, (l - (int64_t)part * 4) * dl, (l - (int64_t)part * 4 + 1) * dl);
                        dlp(k, L_CH_W).addmm_(dP, trk(l, TR_C));                   // dWch_k += dP_kl Wc_l^T
                        dlp(l, L_C_W).addmm_(trk(k, TR_CH), dP);                   // dWc_l  += Wch_k^T dP_kl
                    }
            }
        }

        // ---- backbone on the main stream: video encoder, sentence / word features, the two LSTM layers (models.py:38-83)
        std::vector<Tensor> dbb(P_LAYER0), lstm_bufs;
        // the backbone's weight halves: on the word stream (idle by now) when there is one -- the weight stream still holds layer 0's
        // moment-unit contraction, and these short kernels close the step
        HStream bstr = (tail_split && wordst != curs && wstr != curs) ? wordst : wstr;
        {
            const int Din = i32(st.vx.size(2));
            const int64_t pe_rows = all[P_PE].size(0);
            Tensor dfs_video = at::empty_like(fs);
            dbb[P_VE_W] = at::empty({D, Din}, opt); dbb[P_VE_B] = at::empty({D}, opt);
            dbb[P_PE] = pe_rows != T ? at::zeros({pe_rows, D}, opt) : at::empty({T, D}, opt);
            // every call below is split into its inputs half (main stream, the dependent chain) and its weights half (weight stream); the
            // intermediate of a pair lives in a buffer of its own (the per-stream scratch is reused by the next call on that stream)
            auto own = [&](size_t nbytes) {
                Tensor t = at::empty({(int64_t)nbytes + 256}, at::TensorOptions().dtype(at::kByte).device(dev));
                keep.push_back(t);
                return t;
            };
            Tensor wsv = own(smin_video_encoder_bwd_workspace_bytes(B, Ti, Din, D));
            SMIN_CK(smin_video_encoder_bwd(cur(), fp(df), fp(st.fv), fp(fs), fp(st.vmaskf), fp(st.vx), B, Ti, Din, D, nullptr, nullptr, nullptr, fpm(dfs_video), wsv.data_ptr(),
                                           (size_t)wsv.numel()));
            dfs_parts.push_back(dfs_video);
            if (wordst != curs) await(curs, words_done);
            Tensor dfs_total = sum_list(dfs_parts), dfw_total = sum_list(dfw_parts);
            // f_s = [f_w[b, len_b - 1, :H] | f_w[b, 0, H:]] (models.py:60-62)
            SMIN_CK(smin_sentence_feature_bwd(cur(), fp(dfs_total), ip(st.len32), B, Nq, i32(H), fpm(dfw_total)));
            Tensor dH = Nq_in < Nq ? dfw_total.slice(1, 0, Nq_in).contiguous() : dfw_total;
            keep.push_back(dfs_total); keep.push_back(dfw_total); keep.push_back(dH); keep.push_back(df);
            wait_stream(bstr, curs);
            {
                StreamScope sc(bstr);
                SMIN_CK(smin_video_encoder_bwd(cur(), nullptr, fp(st.fv), fp(fs), fp(st.vmaskf), fp(st.vx), B, Ti, Din, D, fpm(dbb[P_VE_W]), fpm(dbb[P_VE_B]), fpm(dbb[P_PE]),
                                               nullptr, wsv.data_ptr(), (size_t)wsv.numel()));
            }
            for (int layer = 1; layer >= 0; --layer) {
                LstmState& ls = st.lstm[layer];
                const int In = i32(ls.x.size(2)), Hh = i32(H);
                Tensor dX = layer > 0 ? at::empty_like(ls.x) : Tensor();
                Tensor dWih = at::empty_like(ls.Wih), dbias = at::empty({8 * H}, opt), dWhh = at::empty_like(ls.Whh);
                lstm_bufs.push_back(dWih); lstm_bufs.push_back(dbias); lstm_bufs.push_back(dWhh);
                Tensor wsl = own(smin_bilstm_layer_bwd_workspace_bytes(B, i32(Nq_in), In, Hh));
                SMIN_CK(smin_bilstm_layer_bwd(cur(), fp(dH), fp(ls.x), fp(ls.Hout), fp(ls.G), fp(ls.Cs), fp(WihT[layer]), fp(ls.Whh), ip(st.len32), B, i32(Nq_in), In, Hh,
                                              fpm(dX), nullptr, nullptr, nullptr, wsl.data_ptr(), (size_t)wsl.numel()));
                // the weight gradients: three independent pieces; the last layer's (nothing else is left to run by then) on three streams,
                // pieces that share a stream in one call.  b_ih and b_hh get the same gradient in two tensors (dbias2: one tensor handed
                // to both made autograd clone it, a launch per bias at the very end of the step).
                Tensor dbias2 = at::empty({8 * H}, opt);
                lstm_bufs.push_back(dbias2);
                HStream piece[3] = {bstr, bstr, bstr};
                if (layer == 0 && bstr == wordst && tail != curs) { piece[1] = tail; piece[2] = wstr; }
                for (int pc = 0; pc < 3; ++pc) {
                    int which = 1 << pc;
                    while (pc + 1 < 3 && piece[pc + 1] == piece[pc]) which |= 1 << ++pc;
                    wait_stream(piece[pc], curs);
                    StreamScope sc(piece[pc]);
                    SMIN_CK(smin_bilstm_layer_bwd_weights(cur(), which, fp(ls.x), fp(ls.Hout), B, i32(Nq_in), In, Hh, fpm(dWih), fpm(dbias), fpm(dbias2), fpm(dWhh),
                                                          wsl.data_ptr(), (size_t)wsl.numel()));
                }
                if (dX.defined()) keep.push_back(dX);
                const int64_t H4 = 4 * H;
                Tensor* o = &dbb[P_LSTM + 8 * layer];
                o[0] = dWih.slice(0, 0, H4); o[1] = dWhh[0]; o[2] = dbias.slice(0, 0, H4); o[3] = dbias2.slice(0, 0, H4);
                o[4] = dWih.slice(0, H4); o[5] = dWhh[1]; o[6] = dbias.slice(0, H4); o[7] = dbias2.slice(0, H4);
                dH = dX;
            }
        }
        wait_stream(curs, tail);
        wait_stream(curs, wstr);
        wait_stream(curs, wordst);
        if (sync.on) {
            std::vector<Tensor> late{dbb[P_VE_W], dbb[P_VE_B], dbb[P_PE]};
            for (auto& t : lstm_bufs) late.push_back(t);
            sync.reduce(late, curs);
            sync.join(curs);
        }

        variable_list out(N_FIXED + all.size());
        for (size_t i = 0; i < all.size(); ++i) out[N_FIXED + i] = i < (size_t)P_LAYER0 ? dbb[i] : dprm[i - P_LAYER0];
        return out;
    }
};

// ---------------------------------------------------------------- the model

struct Words { Tensor Mq, uq, what, shat; };

std::tuple<Tensor, Tensor, Tensor, Tensor> smin_forward(const Tensor& video_features, const Tensor& video_mask, const Tensor& query_features, const Tensor& query_mask_in,
                                                        const Tensor& length_mask, const Tensor& moment_mask, at::TensorList prm, at::IntArrayRef cfg)
{
    // The reference's dataset pads queries and their mask to max_query_length (dataset.py:35, 173); a batch cut to its longest query is
    // taken too: the word features are padded below as models.py:58-59 does, and the mask here, since every kernel reads max_query_length columns.
    TORCH_CHECK(cfg.size() >= 10 && query_features.dim() == 3, "smin_forward: query_features (B, words, dim) and cfg of at least 10 entries");
    Tensor query_mask = query_mask_in.reshape({query_features.size(0), -1});
    TORCH_CHECK((query_mask.size(1) == query_features.size(1) || query_mask.size(1) == cfg[6]) && query_features.size(1) <= cfg[6], "smin_forward: query_mask has ", query_mask.size(1),
                " columns for ", query_features.size(1), " words (max_query_length ", cfg[6], ")");
    if (query_mask.size(1) < cfg[6]) query_mask = at::constant_pad_nd(query_mask, {0, cfg[6] - query_mask.size(1)}, 0);
    TORCH_CHECK(video_features.is_cuda(), "smin_forward runs on a HIP device only (there is no CPU fallback)");
    TORCH_CHECK(cfg.size() >= 10, "smin_forward: cfg = [T, L, C, D, dl, layers, max_query_length, H, overlap_boundary, overlap_prep(, fused_core, async_weights, bf16_operand_storage, grad_sync, known_cell_count or -1, tail_split)]");
    const int64_t T = cfg[0], L = cfg[1], C = cfg[2], D = cfg[3], nl = cfg[5], maxq = cfg[6], H = cfg[7];
    const bool overlap_boundary = cfg[8] != 0, overlap_prep = cfg[9] != 0;
    TORCH_CHECK((int64_t)prm.size() == P_LAYER0 + nl * L_COUNT + 8, "smin_forward: expected ", P_LAYER0 + nl * L_COUNT + 8, " parameters, got ", prm.size());
    const at::Device dev = video_features.device();
    c10::hip::HIPGuard device_guard(dev.index());
    const int64_t B = video_features.size(0), Tn = video_features.size(1), Nq = query_features.size(1);
    auto lp = [&](int64_t k, int which) -> const Tensor& { return prm[P_LAYER0 + k * L_COUNT + which]; };
    const Tensor* loc = &prm[P_LAYER0 + nl * L_COUNT];
    if (cfg.size() >= 11 && cfg[10] != 0 && !video_features.requires_grad() && !query_features.requires_grad()) {      // the whole model as one node
        const int64_t flags = (overlap_boundary ? SminCore::F_OVERLAP_BOUNDARY : 0) | (overlap_prep ? SminCore::F_OVERLAP_PREP : 0) |
                              ((cfg.size() >= 12 && cfg[11] != 0) ? SminCore::F_ASYNC_WEIGHTS : 0) | ((cfg.size() >= 13 && cfg[12] != 0) ? SminCore::F_BF16_OPERANDS : 0) |
                              ((cfg.size() >= 14 && cfg[13] != 0) ? SminCore::F_GRAD_SYNC : 0) |
                              ((cfg.size() >= 16 && cfg[15] == 0) ? SminCore::F_NO_TAIL_SPLIT : 0) |
                              ((cfg.size() >= 15 && cfg[14] >= 0) ? ((cfg[14] + 1) << 16) : 0);      // cfg[14]: the number of valid cells, when the caller knows it
        auto out = SminCore::apply(video_features, video_mask, query_features, query_mask, length_mask, moment_mask, T, L, C, nl, maxq, H, flags, prm);
        return std::make_tuple(out[0], out[1], out[2], out[3]);
    }

    // ---- layout, part 1: the cell count leaves for the host now and is waited for after the backbone is queued
    Tensor mm = moment_mask.scalar_type() == at::kBool ? moment_mask : moment_mask.ne(0);
    Tensor host_n = at::empty({1}, at::TensorOptions().dtype(at::kLong).pinned_memory(true));
    host_n.copy_(mm.sum().reshape({1}), /*non_blocking=*/true);
    hipEvent_t count_ready = next_event();
    TORCH_CHECK(hipEventRecord(count_ready, c10::hip::getCurrentHIPStream().stream()) == hipSuccess, "hipEventRecord failed");

    // ---- backbone (models.py:38-83): BiLSTM x 2, heads, fused video encoder
    Tensor qm = query_mask.reshape({B, -1});
    Tensor length = qm.sum(1);
    Tensor len32 = length.to(at::kInt);
    Tensor x = query_features;
    for (int layer = 0; layer < 2; ++layer) {
        const Tensor* w = &prm[P_LSTM + 8 * layer];
        x = BiLstmLayer::apply(x, len32, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]);
    }
    Tensor fw = x;
    if (Nq < maxq) fw = at::constant_pad_nd(fw, {0, 0, 0, maxq - Nq}, 0);
    fw = fw.contiguous();
    Tensor last = (length.to(at::kLong) - 1).clamp_min(0).view({B, 1, 1}).expand({B, 1, H});
    Tensor fs = at::cat({fw.slice(2, 0, H).gather(1, last).view({B, H}), fw.select(1, 0).slice(1, H)}, 1);
    Tensor f = VideoFuse::apply(video_features, prm[P_VE_W], prm[P_VE_B], prm[P_PE], fl(video_mask.reshape({B * Tn})), fs);

    // ---- layout, part 2
    TORCH_CHECK(hipEventSynchronize(count_ready) == hipSuccess, "hipEventSynchronize failed");
    Layout lay;
    lay.N = i32(host_n.const_data_ptr<int64_t>()[0]); lay.B = i32(B); lay.L = i32(L);
    {
        auto io = at::TensorOptions().dtype(at::kInt).device(dev);
        Tensor mask8 = mm.contiguous().view(at::kByte);
        lay.cells = at::empty({lay.N, 4}, io); lay.row_ptr = at::empty({B * L + 1}, io); lay.cellmap = at::empty({B, L, L}, io);
        SMIN_CK(smin_build_cells(cur(), static_cast<const uint8_t*>(mask8.const_data_ptr()), lay.B, lay.L, 0, lay.cells.data_ptr<int32_t>(),
                                 lay.row_ptr.data_ptr<int32_t>(), lay.cellmap.data_ptr<int32_t>()));
    }
    const int64_t N = lay.N;
    Tensor qmf = fl(qm), lmf = fl(length_mask);

    // ---- parameter-only work on the second stream (SMIN._forward_stream in modules.py has the commentary)
    HStream curs = c10::hip::getCurrentHIPStream(dev.index());
    HStream side = overlap_boundary ? side_stream(dev.index()) : curs;
    HStream prep = overlap_prep ? side : curs;
    std::vector<Tensor> consts(nl), Pcat_first(nl);
    std::vector<std::vector<Tensor>> Pcats(nl);
    std::vector<Words> words(nl);
    std::vector<std::pair<Tensor, Tensor>> mu_w(nl);
    Tensor Wch_all, const_all;
    wait_stream(prep, curs);
    {
        StreamScope sc(prep);
        Tensor bsum;
        for (int64_t k = 0; k < nl; ++k) {
            consts[k] = bsum.defined() ? lp(k, L_CH_B) + at::mv(lp(k, L_CH_W), bsum) : lp(k, L_CH_B);
            bsum = bsum.defined() ? bsum + lp(k, L_C_B) : lp(k, L_C_B);
        }
        std::vector<Tensor> wch;
        for (int64_t k = 0; k < nl; ++k) wch.push_back(lp(k, L_CH_W));
        Wch_all = at::cat(wch);
        const_all = consts[0];
        // a layer's word-side operands in one launch.  One node per layer, not one for all: a layer's backward then runs as soon
        // as that layer's attention gradients exist instead of after the whole backward pass (where it delayed the query encoder)
        {
            // every layer's word-side operands in one launch (one autograd node: its backward then runs at the tail of the backward
            // pass; one node per layer measured 1.4 ms/step slower -- the three backward launches then run on the second stream
            // beside the layers' matrix kernels and take CUs from the critical path)
            std::vector<Tensor> wp;
            for (int64_t k = 0; k < nl; ++k)
                for (int which : {L_WH_W, L_WH_B, L_SH_W, L_SH_B, L_AK_W, L_AK_B, L_AQ_W, L_AQ_B}) wp.push_back(lp(k, which));
            auto wo = WordPrep::apply(fw, fs, qmf, at::TensorList(wp));
            for (int64_t k = 0; k < nl; ++k) { words[k].what = wo[4 * k]; words[k].shat = wo[4 * k + 1]; words[k].Mq = wo[4 * k + 2]; words[k].uq = wo[4 * k + 3]; }
        }
        for (int64_t k = 0; k < nl; ++k) {
            for (int64_t lo = 0; lo < k; lo += 4) {
                std::vector<Tensor> parts;
                for (int64_t l = lo; l < std::min(lo + 4, k); ++l) parts.push_back(at::matmul(lp(k, L_CH_W), lp(l, L_C_W)));
                Pcats[k].push_back(at::cat(parts, 1));
            }
            mu_w[k] = {at::cat({lp(k, L_FB_W).view({D, D}), lp(k, L_FC_W).view({D, D})}, 1), lp(k, L_FB_B) + lp(k, L_FC_B)};
        }
    }
    wait_stream(curs, prep);
    if (prep != curs) {
        record_stream(Wch_all, curs);
        for (int64_t k = 0; k < nl; ++k) {
            record_stream(consts[k], curs);
            record_stream(words[k].Mq, curs); record_stream(words[k].uq, curs); record_stream(words[k].what, curs); record_stream(words[k].shat, curs);
            for (auto& p : Pcats[k]) record_stream(p, curs);
            record_stream(mu_w[k].first, curs); record_stream(mu_w[k].second, curs);
        }
    }
    if (side != curs) {
        for (const Tensor* t : {&fw, &fs, &qmf, &lmf, &lay.cells, &lay.row_ptr, &lay.cellmap}) record_stream(*t, side);
    }

    auto pmv = ProposalMeans::apply(f, lay.cells, lay.row_ptr, lay.cellmap, N, T, L, C);
    Tensor fm = pmv[0], fb = pmv[1];
    const std::optional<Tensor> none;
    Tensor g_all = LinearRows::apply(Wch_all, none, none, none, 1, at::TensorList{f.reshape({-1, D})});
    auto pgs = ClipWindowMeans::apply(g_all.view({B, T, -1}), const_all, lay.cells, lay.row_ptr, lay.cellmap, N, T, L, C, nl);

    Tensor cumean, Hs, Hsum;
    std::vector<Tensor> hist;
    for (int64_t k = 0; k < nl; ++k) {
        const bool lastl = k == nl - 1;
        const int64_t n_hbar = lastl ? 2 : ((k > 0 || nl < 3) ? 3 : 4);
        auto views = Gate::apply(fm, fs, lay.cells, lay.row_ptr, B, L, n_hbar, k == 0 ? 2 : 1);
        Tensor hbar_c = views[0], hbar_b = views[1], fm_res = views[n_hbar];
        if (k == 0) cumean = views[n_hbar + 1];                          // mean_c f_c of the proposal map is f_m
        // boundary unit on the second stream beside the content stream; joins before the moment unit
        Tensor bu;
        wait_stream(side, curs);
        {
            StreamScope sc(side);
            bu = BoundaryUnitFn::apply(fb, fw, fs, hbar_b, lp(k, L_BQ_W), lp(k, L_BQ_B), lp(k, L_BK_W), lp(k, L_BK_B), qmf, lmf, lay.cells, lay.row_ptr, N);
        }
        if (side != curs) { record_stream(fb, side); record_stream(hbar_b, side); }
        const Tensor& Wch = lp(k, L_CH_W);
        Tensor chat = pgs[k];
        for (size_t part = 0, lo = 0; lo < hist.size(); ++part, lo += 4) {
            std::optional<Tensor> hp;
            if (lo == 0) hp = LinearRows::apply(Wch, none, none, none, 1, at::TensorList{Hs});
            std::vector<Tensor> xs(hist.begin() + lo, hist.begin() + std::min(lo + 4, hist.size()));
            chat = LinearRows::apply(Pcats[k][part], lo == 0 ? std::optional<Tensor>(consts[k]) : none, std::optional<Tensor>(chat), hp, C, at::TensorList(xs));
        }
        const Words& w = words[k];
        auto ca = ContentAttn::apply(chat, w.Mq, w.uq, w.what, w.shat, qmf, lay.cells, lay.row_ptr, N, L, C, !lastl);
        Tensor cc = ca[0], ccmean = ca[1];
        cumean = LinearRows::apply(lp(k, L_C_W), std::optional<Tensor>(lp(k, L_C_B)), std::optional<Tensor>(cumean), std::optional<Tensor>(hbar_c), 1, at::TensorList{ccmean});
        if (!lastl) {
            if (!Hs.defined()) { Hs = views[2]; Hsum = n_hbar > 3 ? views[3] : views[2]; }
            else { Hs = Hsum + views[2]; Hsum = Hs; }
            hist.push_back(cc);
        }
        wait_stream(curs, side);
        if (side != curs) record_stream(bu, curs);
        auto mu = MomentUnitFn::apply(cumean, fm_res, bu, mu_w[k].first, mu_w[k].second, lay.cells, lay.row_ptr, lay.cellmap);
        fm = mu[0]; cumean = mu[1];
        fb = bu;
    }
    // Localization (models.py:335-344)
    Tensor wb = at::stack({loc[2].view({D}), loc[4].view({D}), loc[6].view({D})});
    Tensor bb = at::cat({loc[3], loc[5], loc[7]});
    auto sc = ScoreMap::apply(fm, fb, loc[0].view({D}), loc[1], wb, bb, lmf, lay.cells);
    Tensor psea = sc[1];
    return std::make_tuple(sc[0], psea[0], psea[1], psea[2]);
}

Tensor smin_loss(const Tensor& pm, const Tensor& ym, const Tensor& sm, const Tensor& moment_mask, const Tensor& ps, const Tensor& ys, const Tensor& ss, const Tensor& pe,
                 const Tensor& ye, const Tensor& se, const Tensor& pa, const Tensor& ya, const Tensor& length_mask)
{
    TORCH_CHECK(pm.is_cuda(), "smin_loss runs on a HIP device only (there is no CPU fallback)");
    c10::hip::HIPGuard device_guard(pm.device().index());
    return LossNode::apply(pm, ps, pe, pa, ym, sm, moment_mask, ys, ss, ye, se, ya, length_mask);
}

}  // namespace

TORCH_LIBRARY(smin_hip, m)
{
    // SMIN.forward (reference models.py:367-377): the six forward arguments, the parameters in SMIN._native_params order and
    // cfg = [T, L, C, D, dl, num_smi_layers, max_query_length, lstm_hidden_size, overlap_boundary, overlap_prep, fused_core]
    m.def("smin_forward(Tensor video_features, Tensor video_mask, Tensor query_features, Tensor query_mask, Tensor length_mask, Tensor moment_mask, "
          "Tensor[] params, int[] cfg) -> (Tensor, Tensor, Tensor, Tensor)", &smin_forward);
    // restated loss_fn of the reference's train loop (main.py:110-116), same argument order
    m.def("smin_loss(Tensor pm, Tensor ym, Tensor sm, Tensor moment_mask, Tensor ps, Tensor ys, Tensor ss, Tensor pe, Tensor ye, Tensor se, Tensor pa, Tensor ya, "
          "Tensor length_mask) -> Tensor", &smin_loss);
    m.def("abi_version() -> int", []() -> int64_t { return smin_abi_version(); });
    // the status word of smin_build_cells_n on a device (non-zero after a step whose cfg[14] cell count did not match its mask)
    m.def("layout_status(Device device) -> Tensor", [](c10::Device dev) { return layout_status(dev); });
    // data parallel: the process group (c10d group name) the one-node backward averages its gradients over, see GradSync;
    // cfg[13] of smin_forward switches the exchange on per call.  coalesced_avg: the backend takes grouped "avg" all-reduces (RCCL)
    m.def("set_grad_sync(str group_name, int world, bool coalesced_avg) -> ()", [](std::string group, int64_t world, bool coalesced_avg) {
        GradSyncConfig& c = grad_sync_config();
        c.group = std::move(group); c.world = (int)world; c.coalesced_avg = coalesced_avg;
    });
}
