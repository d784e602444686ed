// Host-side autograd node of the batch-1 path in C++ (built with g++ against libtorch; no device code here).
//
// trackmpnn_amd/small.py implements the same node in Python (`_SmallIter`); at batch 1 the Python bookkeeping of a
// forward + backward pair (~100 us: Function.apply, a dozen tensor allocations, two ctypes calls) exceeds the GPU time
// of the four launches it enqueues.  This node does the identical sequence -- allocate, call the C ABI
// (tmpnn_mp_iter_fwd / tmpnn_mp_iter_bwd through function pointers handed over from Python), keep what the backward
// needs -- without the interpreter.  Two gradient modes: IN PLACE (GradBucket: parameter gradients go straight into
// p.grad, the third input is a dummy `anchor` that only tells autograd the outputs need a backward) and SINK (default
// autograd semantics: the third input is the model's gradient sink, a tensor whose VALUES nobody reads and whose
// gradient is all parameter gradients of this call in one flat buffer; trackmpnn_amd/small.py:_ParamSink hands the
// slices to the parameters, so autograd accumulates one tensor per call instead of one per parameter per call).
// torch is plumbing: memory, autograd edges, nothing else.
#include <torch/extension.h>

#include <cstdint>
#include <chrono>
#include <cstring>
#include <tuple>

#include "tmpnn.h"

namespace {

using fwd_fn = int (*)(const tmpnn_mp_params*, const float*, const tmpnn_dgraph*, int, const float*, int, float*, int,
                       float*, float*, float*, float*, size_t, tmpnn_stream);
using bwd_fn = int (*)(const tmpnn_mp_params*, const float*, const tmpnn_dgraph*, int, const float*, int, const float*,
                       const float*, const float*, const float*, int, const float*, int, const float*, int, const float*,
                       float*, float*, const tmpnn_mp_params*, void*, size_t, tmpnn_stream);
using err_fn = const char* (*)(void);
using bind_fn = int (*)(void*, int, int, tmpnn_dgraph*);

struct CallInfo {
    int64_t f_fwd, f_bwd, f_err;   // function addresses in libtmpnn.so
    int64_t params, grads;         // tmpnn_mp_params* (parameters / gradient buffers), owned by Python
    int64_t prep;                  // float* operand images
    int64_t f_bind;                // tmpnn_dgraph_bind
    int64_t N, G, H, IN_e, F_total;
    int64_t stream;
    int64_t spare;                 // spare rows behind h_out
    bool training, need_grad, append;
};

inline size_t save_floats(int64_t N, int64_t n, int64_t G, int64_t H) {
    return (size_t)(G * 4 * N * H + G * N * H + G * (n > 0 ? n : 1) * H + 2 * G * H + G * (n + 1) + 4);
}
// gradient struct of the SINK mode: `tmpl` holds 1 + byte offset into the flat buffer where a pointer goes (0 = NULL)
inline void rebase(float*& p, char* base) {
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    p = v ? reinterpret_cast<float*>(base + (v - 1)) : nullptr;
}
inline tmpnn_mp_params rebased(const tmpnn_mp_params* tmpl, void* flat) {
    tmpnn_mp_params s = *tmpl;
    char* b = static_cast<char*>(flat);
    for (int g = 0; g < 3; ++g) {
        rebase(s.w1[g], b); rebase(s.b1[g], b); rebase(s.gamma[g], b); rebase(s.beta[g], b); rebase(s.w2[g], b); rebase(s.b2[g], b);
        s.run_mean[g] = nullptr; s.run_var[g] = nullptr; s.num_batches_tracked[g] = nullptr;
        rebase(s.e_wih[g], b); rebase(s.e_whh[g], b); rebase(s.e_bih[g], b); rebase(s.e_bhh[g], b);
        rebase(s.n_wih[g], b); rebase(s.n_whh[g], b); rebase(s.n_bih[g], b); rebase(s.n_bhh[g], b);
    }
    rebase(s.w_node, b); rebase(s.b_node, b); rebase(s.w_edge, b); rebase(s.b_edge, b);
    return s;
}
inline size_t bwd_ws_bytes(int64_t N, int64_t n, int64_t G, int64_t H, int64_t IN_e) {
    int64_t nb = (N + 15) / 16 + 2;
    if (nb > 96) nb = 96;
    if (nb < 2) nb = 2;
    const int64_t slab = 3 * H * (IN_e + H) + 7 * H + 4;
    return (size_t)(4 * (N * G * IN_e + nb * G * slab + G * 2 * n * H + 16));
}

class SmallIterFn : public torch::autograd::Function<SmallIterFn> {
   public:
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor x,
                                                  c10::optional<torch::Tensor> h_in, torch::Tensor anchor,
                                                  torch::Tensor arena, std::vector<int64_t> info,
                                                  std::vector<torch::Tensor> keep) {
        (void)anchor;
        CallInfo ci;
        TORCH_CHECK(info.size() == 18, "fast_iter: bad call descriptor");
        ci.f_fwd = info[0]; ci.f_bwd = info[1]; ci.f_err = info[2]; ci.params = info[3]; ci.grads = info[4];
        ci.prep = info[5]; ci.f_bind = info[6]; ci.N = info[7]; ci.G = info[8]; ci.H = info[9]; ci.IN_e = info[10];
        ci.F_total = info[11]; ci.stream = info[12]; ci.spare = info[13];
        ci.training = info[14] != 0; ci.need_grad = info[15] != 0; ci.append = info[16] != 0;
        const int64_t N = ci.N, GH = ci.G * ci.H, n = x.size(0), N_old = N - n;
        const bool has_h = h_in.has_value() && h_in->defined();
        if (!has_h) {
            TORCH_CHECK_VALUE(N_old == 0, "h_in is None but the graph has ", N_old, " rows that are not new");
        } else {
            TORCH_CHECK_VALUE(h_in->size(0) == N_old && h_in->size(1) == GH, "h_in must be [", N_old, ", ", GH, "] (N - n, G*H)");
        }
        if (n > 0) { TORCH_CHECK_VALUE(x.size(1) == ci.F_total, "x must be [", n, ", ", ci.F_total, "]"); }
        TORCH_CHECK_VALUE(!(ci.training && n == 1), "Expected more than 1 value per channel when training, got input size [1, ", ci.H, "]");
        tmpnn_dgraph dg;       // the graph's arrays live in `arena` (kept alive by this node); the struct is rebuilt from it
        TORCH_CHECK(reinterpret_cast<bind_fn>(ci.f_bind)(arena.data_ptr(), (int)N, (int)N, &dg) == 0, "tmpnn_dgraph_bind failed");
        auto opts = x.options().dtype(torch::kFloat32).requires_grad(false);
        torch::Tensor xd = x.detach();
        if (xd.scalar_type() != torch::kFloat32 || !xd.is_contiguous()) xd = xd.to(torch::kFloat32).contiguous();
        // the state the iteration reads
        torch::Tensor h_cat;
        if (ci.append && has_h) {
            torch::Tensor hd = h_in->detach();
            if (hd.is_contiguous() && hd.scalar_type() == torch::kFloat32)
                h_cat = at::empty({0}, opts).set_(hd.storage(), hd.storage_offset(), {N, GH}, {GH, 1});
        }
        if (!h_cat.defined()) {
            if (has_h && n == 0) {
                torch::Tensor hd = h_in->detach();
                h_cat = (hd.is_contiguous() && hd.scalar_type() == torch::kFloat32) ? hd : hd.to(torch::kFloat32).contiguous();
            } else {
                h_cat = at::empty({N, GH}, opts);
                if (N_old > 0) h_cat.narrow(0, 0, N_old).copy_(h_in->detach());
            }
        }
        torch::Tensor buf = at::empty({(N + ci.spare) * GH}, opts);
        torch::Tensor h_out = at::empty({0}, opts).set_(buf.storage(), 0, {N, GH}, {GH, 1});
        torch::Tensor logits = at::empty({N, 1}, opts), scores = at::empty({N, 1}, opts);
        torch::Tensor save;
        size_t nsave = 0;
        if (ci.need_grad || n > 0) {
            nsave = save_floats(N, n, ci.G, ci.H);
            save = at::empty({(int64_t)nsave}, opts);
        }
        const int rc = reinterpret_cast<fwd_fn>(ci.f_fwd)(
            reinterpret_cast<const tmpnn_mp_params*>(ci.params), reinterpret_cast<const float*>(ci.prep),
            &dg, (int)n, n > 0 ? xd.data_ptr<float>() : nullptr,
            n > 0 ? (int)xd.size(1) : 0, h_cat.data_ptr<float>(), ci.training ? 1 : 0, h_out.data_ptr<float>(),
            logits.data_ptr<float>(), scores.data_ptr<float>(), save.defined() ? save.data_ptr<float>() : nullptr, nsave,
            reinterpret_cast<tmpnn_stream>(ci.stream));
        TORCH_CHECK(rc == 0, "tmpnn_mp_iter_fwd failed (code ", rc, "): ", reinterpret_cast<err_fn>(ci.f_err)());
        if (ci.need_grad) {
            ctx->saved_data["info"] = info;
            // What the backward dereferences must live as long as this node, not as long as the Python caches that built
            // it: the two parameter structs are copied by value, the operand images and (in-place gradient mode) the
            // gradient tensors are held by reference.
            auto pod = [](int64_t addr) {
                torch::Tensor t = at::empty({(int64_t)sizeof(tmpnn_mp_params)}, at::TensorOptions().dtype(torch::kByte));
                std::memcpy(t.data_ptr(), reinterpret_cast<const void*>(addr), sizeof(tmpnn_mp_params));
                return t;
            };
            ctx->saved_data["params_pod"] = pod(ci.params);
            ctx->saved_data["grads_pod"] = pod(ci.grads);
            ctx->saved_data["keep"] = keep;
            ctx->saved_data["n"] = n;
            ctx->saved_data["has_h"] = has_h;
            ctx->saved_data["x_needs"] = x.requires_grad();
            // plain (non-variable) tensors kept alive by the node: they are internal buffers, not autograd values
            ctx->saved_data["xd"] = xd;
            ctx->saved_data["h_cat"] = h_cat;
            ctx->saved_data["save"] = save;
            ctx->saved_data["arena"] = arena;
            ctx->save_for_backward({scores});          // an OUTPUT: saved through the autograd mechanism (no cycle)
        }
        ctx->set_materialize_grads(false);
        return {scores, logits, h_out};
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                   torch::autograd::variable_list grad_outputs) {
        TORCH_CHECK(ctx->saved_data.count("info") != 0,
                    "fast_iter: backward through a call that saved nothing (it was recorded with need_grad = false)");
        auto info = ctx->saved_data["info"].toIntVector();
        torch::Tensor params_pod = ctx->saved_data["params_pod"].toTensor();
        torch::Tensor grads_pod = ctx->saved_data["grads_pod"].toTensor();
        const int64_t n = ctx->saved_data["n"].toInt();
        const bool has_h = ctx->saved_data["has_h"].toBool();
        const bool x_needs = ctx->saved_data["x_needs"].toBool();
        torch::Tensor xd = ctx->saved_data["xd"].toTensor();
        torch::Tensor h_cat = ctx->saved_data["h_cat"].toTensor();
        torch::Tensor scores = ctx->get_saved_variables()[0];
        torch::Tensor save = ctx->saved_data["save"].toTensor();
        torch::Tensor arena = ctx->saved_data["arena"].toTensor();
        const int64_t N = info[7], G = info[8], H = info[9], IN_e = info[10], F_total = info[11];
        const int64_t GH = G * H;
        auto opts = h_cat.options();
        tmpnn_dgraph dg;
        TORCH_CHECK(reinterpret_cast<bind_fn>(info[6])(arena.data_ptr(), (int)N, (int)N, &dg) == 0, "tmpnn_dgraph_bind failed");
        auto strided = [](const torch::Tensor& t, torch::Tensor& keep, int& st) -> const float* {
            if (!t.defined()) { st = 0; return nullptr; }
            keep = t.scalar_type() == torch::kFloat32 ? t : t.to(torch::kFloat32);
            int64_t s = keep.numel() > 1 ? keep.stride(0) : 1;
            if (s < 0) { keep = keep.contiguous(); s = 1; }
            st = (int)s;
            return keep.data_ptr<float>();
        };
        torch::Tensor k_ds, k_dl, k_dh;
        int st_ds = 0, st_dl = 0;
        const float* ds = strided(grad_outputs[0], k_ds, st_ds);
        const float* dl = strided(grad_outputs[1], k_dl, st_dl);
        const float* dh = nullptr;
        if (grad_outputs[2].defined()) {
            k_dh = (grad_outputs[2].scalar_type() == torch::kFloat32 && grad_outputs[2].is_contiguous())
                       ? grad_outputs[2] : grad_outputs[2].to(torch::kFloat32).contiguous();
            dh = k_dh.data_ptr<float>();
        }
        torch::Tensor d_h = at::empty({N, GH}, opts);
        torch::Tensor d_x;
        const bool need_x = x_needs && n > 0;
        if (need_x) d_x = at::empty({n, F_total}, opts);
        const size_t wsb = bwd_ws_bytes(N, n, G, H, IN_e);
        torch::Tensor ws = at::empty({(int64_t)(wsb / 4 + 4)}, opts);
        // SINK mode: this call's parameter gradients land in one fresh flat buffer, returned as the sink's gradient
        const int64_t sink_total = info[17];
        torch::Tensor gflat;
        tmpnn_mp_params gs;
        const tmpnn_mp_params* grads = reinterpret_cast<const tmpnn_mp_params*>(grads_pod.data_ptr());
        if (sink_total > 0) {
            gflat = at::zeros({sink_total}, opts);
            gs = rebased(grads, gflat.data_ptr());
            grads = &gs;
        }
        const int rc = reinterpret_cast<bwd_fn>(info[1])(
            reinterpret_cast<const tmpnn_mp_params*>(params_pod.data_ptr()), reinterpret_cast<const float*>(info[5]),
            &dg, (int)n, n > 0 ? xd.data_ptr<float>() : nullptr,
            n > 0 ? (int)xd.size(1) : 0, h_cat.data_ptr<float>(), nullptr, scores.data_ptr<float>(),
            save.data_ptr<float>(), info[14] != 0 ? 1 : 0, ds, st_ds, dl, st_dl, dh, d_h.data_ptr<float>(),
            need_x ? d_x.data_ptr<float>() : nullptr, grads, ws.data_ptr<float>(),
            wsb, reinterpret_cast<tmpnn_stream>(info[12]));
        TORCH_CHECK(rc == 0, "tmpnn_mp_iter_bwd failed (code ", rc, "): ", reinterpret_cast<err_fn>(info[2])());
        if (x_needs && !need_x) d_x = at::zeros({n, F_total}, opts);
        torch::Tensor d_h_in;
        if (has_h && N - n > 0) d_h_in = d_h.narrow(0, 0, N - n);
        return {d_x, d_h_in, gflat, torch::Tensor(), torch::Tensor(), torch::Tensor()};
    }
};

// keep: tensors the backward reads through raw pointers of `info` (operand images; in-place mode: the gradient tensors)
std::vector<torch::Tensor> small_iter(torch::Tensor x, c10::optional<torch::Tensor> h_in, torch::Tensor anchor,
                                      torch::Tensor arena, std::vector<int64_t> info, std::vector<torch::Tensor> keep) {
    return SmallIterFn::apply(x, h_in, anchor, arena, info, keep);
}

// ------------------------------------------------------------------------------------------------------------------------
// One inference timestep of infer.py:70-87 (greedy or, since round 5, --hungarian association) without the interpreter between its launches: update_graph's block append
// (tmpnn_track_extend; the active set of this timestep was derived by the previous step's tmpnn_track_retire), the model call
// in eval mode (tmpnn_mp_iter_fwd) and decode_tracks (tmpnn_track_retire, with the NEXT timestep's active set), then the one
// host read of the timestep.  trackmpnn_amd/loops.py drives it for models on the fused path and graphs of LDS size; every other
// case (Hungarian matching on the host, attention heads, wide cells, re-initialisation, empty timesteps) takes TrackGraph.update / .decode.
// Same kernels, same arguments: the results are those of the Python path bit for bit.
using extend_fn = int (*)(int, int, int, const int32_t*, const int32_t*, int, const int32_t*, const tmpnn_track_rows*, const float*,
                          int, int, float*, int, const tmpnn_dgraph*, void*, size_t, tmpnn_stream);
using retire_fn = int (*)(const tmpnn_dgraph*, const tmpnn_track_rows*, const float*, int, int, int, int32_t*, int, int32_t*, void*,
                          size_t, int32_t*, int32_t*, const tmpnn_track_rows*, const float*, int, int, float*, int, float*, int,
                          int32_t*, int32_t*, tmpnn_stream);
using ints_fn = size_t (*)(int);
using extend_tf_fn = int (*)(int, int, int, const int32_t*, const int32_t*, int, const int32_t*, const tmpnn_track_rows*, const float*,
                             int, const tmpnn_mp_params*, float*, float*, size_t, const tmpnn_dgraph*, const int32_t*, const float*, int,
                             const int32_t*, tmpnn_stream);
using fwd_parts_fn = int (*)(const tmpnn_mp_params*, const float*, const tmpnn_dgraph*, int, const float*, int, float*, int, float*,
                             float*, float*, float*, size_t, int, tmpnn_stream);

// The driver's state and the two halves of a timestep.  FRONT: the block append + index form + input transform
// (tmpnn_track_extend_tf; or tmpnn_track_extend when the one-launch form is switched off).  BACK: the iteration
// (tmpnn_mp_iter_fwd_parts) and decode_tracks (tmpnn_track_retire, with the NEXT timestep's active set).
namespace {
struct Driver {
    extend_fn f_extend; retire_fn f_retire; ints_fn f_ints; extend_tf_fn f_extend_tf; fwd_parts_fn f_fwd_parts; fwd_fn f_fwd;
    err_fn f_err; bind_fn f_bind;
    int32_t* active; const int32_t* track; const float* X; int F; int32_t* y_track; int ND; int32_t* pos_of_det; int32_t* keep_rows;
    int32_t* small; tmpnn_stream stream; int associate; void* hung_ws; size_t hung_ws_bytes; int32_t* notify; int score_rule;
    int ret_win; bool one_launch;
    const tmpnn_mp_params* P; const float* prep; int64_t G, H, GH;
    at::TensorOptions opts, iopts;
};
struct Front {                // what the first launch of a timestep produced / was given
    bool valid = false;
    int64_t cap = 0;          // rows the index form's arena is laid out for (>= the timestep's rows)
    torch::Tensor arena, h_cat, save, feats;
};
}  // namespace

static Driver make_driver(const std::vector<int64_t>& ti, const std::vector<int64_t>& info, const torch::Tensor& h) {
    TORCH_CHECK(ti.size() == 31 && info.size() == 18, "greedy_run: bad descriptors");
    Driver d;
    d.f_extend = reinterpret_cast<extend_fn>(ti[0]);
    d.f_retire = reinterpret_cast<retire_fn>(ti[1]);
    d.f_ints = reinterpret_cast<ints_fn>(ti[2]);
    d.ret_win = (int)ti[8];
    d.active = reinterpret_cast<int32_t*>(ti[10]);
    d.track = reinterpret_cast<const int32_t*>(ti[12]);
    d.X = reinterpret_cast<const float*>(ti[15]);
    d.F = (int)ti[16];
    d.y_track = reinterpret_cast<int32_t*>(ti[17]);
    d.ND = (int)ti[18];
    d.pos_of_det = reinterpret_cast<int32_t*>(ti[19]);
    d.keep_rows = reinterpret_cast<int32_t*>(ti[20]);
    d.small = reinterpret_cast<int32_t*>(ti[21]);
    d.stream = reinterpret_cast<tmpnn_stream>(ti[23]);
    // association rule of decode_tracks: 1 greedy, 2 optimal assignment on the device (--hungarian; its cost scratch, and the
    // next timestep's active set then comes from the launch's second sweep over the rows that stay)
    d.associate = (int)ti[24];
    d.hung_ws = reinterpret_cast<void*>(ti[25]);
    d.hung_ws_bytes = (size_t)ti[26];
    // the counters' mirror in pinned, device-mapped host memory (tmpnn_track_retire `notify`; 0: read `small` back by a copy)
    d.notify = reinterpret_cast<int32_t*>(ti[27]);
    // the block append and the model call's input transform in one launch (tmpnn_track_extend_tf) + the iteration alone
    // (tmpnn_mp_iter_fwd_parts, parts = 1); 0: the two calls as the Python path makes them (tmpnn_track_extend, tmpnn_mp_iter_fwd)
    d.f_extend_tf = reinterpret_cast<extend_tf_fn>(ti[28]);
    d.f_fwd_parts = reinterpret_cast<fwd_parts_fn>(ti[29]);
    d.one_launch = d.f_extend_tf != nullptr && d.f_fwd_parts != nullptr;
    // 8: the model has no TP classifier -- every detection's score is 1 (tmpnn_mp_iter_fwd_parts bit 3; infer.py:77-80)
    d.score_rule = (int)ti[30];
    TORCH_CHECK((d.score_rule & ~8) == 0 && (d.score_rule == 0 || d.f_fwd_parts != nullptr), "greedy_run: score rule ", d.score_rule);
    TORCH_CHECK(d.associate == 1 || (d.associate == 2 && d.hung_ws && d.hung_ws_bytes > 0), "greedy_run: association rule ", d.associate);
    d.f_fwd = reinterpret_cast<fwd_fn>(info[0]);
    d.f_err = reinterpret_cast<err_fn>(info[2]);
    d.f_bind = reinterpret_cast<bind_fn>(info[6]);
    d.P = reinterpret_cast<const tmpnn_mp_params*>(info[3]);
    d.prep = reinterpret_cast<const float*>(info[5]);
    d.G = info[8]; d.H = info[9]; d.GH = d.G * d.H;
    TORCH_CHECK(d.F == info[11], "greedy_run: F = ", d.F);
    d.opts = h.options().requires_grad(false);
    d.iopts = d.opts.dtype(torch::kInt32);
    return d;
}

// FRONT of a timestep.  counts == nullptr: N / A are exact and h [N][GH] is the carried state (extended in place where its
// storage has the room, else copied).  counts != nullptr (the launch is enqueued BEHIND the previous timestep's decode, before its
// counters are known): N / A are upper bounds, the launch reads the exact values on the device, and hbuf is the buffer the
// previous decode compacts the state into (sized by the caller for N + A*D + D rows).
// gather_src (early fronts only): the decode in front left the kept rows' state in place (h_out of its timestep); this launch moves it.
static Front launch_front(const Driver& d, int64_t N, int64_t A, int64_t D, int t, const int32_t* new_ids,
                          const tmpnn_track_rows* rows, const torch::Tensor& h, int64_t cap_rows, const int32_t* counts,
                          const float* gather_src = nullptr) {
    Front f;
    const int64_t n_new = A * D + D, Nt = N + n_new;
    f.cap = Nt;
    f.arena = at::empty({(int64_t)d.f_ints((int)Nt)}, d.iopts);
    tmpnn_dgraph dg;
    TORCH_CHECK(d.f_bind(f.arena.data_ptr(), (int)Nt, (int)Nt, &dg) == 0, "tmpnn_dgraph_bind failed");
    if (counts != nullptr || (cap_rows >= Nt && (int64_t)h.storage().nbytes() >=
                                                    (int64_t)((h.storage_offset() + Nt * d.GH) * sizeof(float)))) {
        TORCH_CHECK((int64_t)h.storage().nbytes() >= (int64_t)((h.storage_offset() + Nt * d.GH) * sizeof(float)),
                    "greedy_run: the state buffer is too small for the enqueued timestep");
        f.h_cat = at::empty({0}, d.opts).set_(h.storage(), h.storage_offset(), {Nt, d.GH}, {d.GH, 1});
    } else {
        f.h_cat = at::empty({Nt, d.GH}, d.opts);
        if (N > 0) f.h_cat.narrow(0, 0, N).copy_(h);
    }
    const size_t nsave = save_floats(Nt, n_new, d.G, d.H);
    f.save = at::empty({(int64_t)nsave}, d.opts);
    int rc;
    if (d.one_launch) {
        rc = d.f_extend_tf((int)N, (int)A, (int)D, d.active, new_ids, t, d.track, rows, d.X, d.F, d.P, f.h_cat.data_ptr<float>(),
                           f.save.data_ptr<float>(), nsave, &dg, counts, gather_src, (int)d.GH, gather_src ? d.keep_rows : nullptr,
                           d.stream);
        TORCH_CHECK(rc == 0, "tmpnn_track_extend_tf failed (code ", rc, "): ", d.f_err());
    } else {
        TORCH_CHECK(counts == nullptr, "greedy_run: an early front needs the one-launch form");
        f.feats = at::empty({n_new, d.F}, d.opts);
        rc = d.f_extend((int)N, (int)A, (int)D, d.active, new_ids, t, d.track, rows, d.X, d.F, d.F, f.feats.data_ptr<float>(), d.F, &dg,
                        nullptr, 0, d.stream);
        TORCH_CHECK(rc == 0, "tmpnn_track_extend failed (code ", rc, "): ", d.f_err());
    }
    f.valid = true;
    return f;
}

// BACK of a timestep on the exact row counts: the iteration, then decode_tracks + the next timestep's active set; the compacted
// state lands in `hbuf` ((Nt + spare) rows: room for the next block, whose front may already be enqueued behind this).
struct Back { torch::Tensor hbuf, h_new, s_new, h_out, logits, scores; };
// defer_gather: the kept rows' state stays in h_out; the NEXT timestep's (early) front moves it into h_new.
static Back launch_back(const Driver& d, Front& f, int64_t N, int64_t A, int64_t D, int t_upto, int next_t,
                        const tmpnn_track_rows* rows_cur, const tmpnn_track_rows* rows_out, int64_t spare, bool defer_gather) {
    Back b;
    const int64_t n_new = A * D + D, Nt = N + n_new;
    tmpnn_dgraph dg;
    TORCH_CHECK(Nt <= f.cap && d.f_bind(f.arena.data_ptr(), (int)f.cap, (int)Nt, &dg) == 0, "tmpnn_dgraph_bind failed");
    torch::Tensor h_cat = f.h_cat.size(0) == Nt ? f.h_cat
                                                 : at::empty({0}, d.opts).set_(f.h_cat.storage(), f.h_cat.storage_offset(), {Nt, d.GH}, {d.GH, 1});
    b.h_out = at::empty({Nt, d.GH}, d.opts); b.logits = at::empty({Nt, 1}, d.opts); b.scores = at::empty({Nt, 1}, d.opts);
    int rc;
    if (d.one_launch)
        rc = d.f_fwd_parts(d.P, d.prep, &dg, (int)n_new, nullptr, 0, h_cat.data_ptr<float>(), 0, b.h_out.data_ptr<float>(),
                           b.logits.data_ptr<float>(), b.scores.data_ptr<float>(), nullptr, 0, 1 | d.score_rule, d.stream);
    else if (d.f_fwd_parts != nullptr)
        rc = d.f_fwd_parts(d.P, d.prep, &dg, (int)n_new, f.feats.data_ptr<float>(), d.F, h_cat.data_ptr<float>(), 0,
                           b.h_out.data_ptr<float>(), b.logits.data_ptr<float>(), b.scores.data_ptr<float>(), f.save.data_ptr<float>(),
                           (size_t)f.save.numel(), d.score_rule, d.stream);
    else
        rc = d.f_fwd(d.P, d.prep, &dg, (int)n_new, f.feats.data_ptr<float>(), d.F, h_cat.data_ptr<float>(), 0, b.h_out.data_ptr<float>(),
                     b.logits.data_ptr<float>(), b.scores.data_ptr<float>(), f.save.data_ptr<float>(), (size_t)f.save.numel(), d.stream);
    TORCH_CHECK(rc == 0, "tmpnn_mp_iter_fwd failed (code ", rc, "): ", d.f_err());
    b.hbuf = at::empty({(Nt + spare) * d.GH}, d.opts);
    b.h_new = at::empty({0}, d.opts).set_(b.hbuf.storage(), 0, {Nt, d.GH}, {d.GH, 1});
    b.s_new = at::empty({Nt, 1}, d.opts);
    if (d.notify != nullptr) __atomic_store_n(d.notify + 4, 0, __ATOMIC_RELAXED);      // (the mirror's flag: armed per launch)
    rc = d.f_retire(&dg, rows_cur, b.scores.data_ptr<float>(), d.associate, t_upto, d.ret_win, d.y_track, d.ND, d.pos_of_det,
                    d.associate == 2 ? d.hung_ws : nullptr, d.associate == 2 ? d.hung_ws_bytes : 0, d.keep_rows, d.small, rows_out,
                    b.h_out.data_ptr<float>(), (int)d.GH, (int)d.GH, defer_gather ? nullptr : b.h_new.data_ptr<float>(), (int)d.GH,
                    b.s_new.data_ptr<float>(), next_t, d.active, d.notify, d.stream);
    TORCH_CHECK(rc == 0, "tmpnn_track_retire failed (code ", rc, "): ", d.f_err());
    return b;
}

// the one host read of a timestep: kept rows, status, kept det rows, the next active-set size.  With a mirror: poll its flag (the
// launch stores the counters there, then the flag) -- no copy is enqueued and the counts are here while the kept rows' state is
// still moving; a flag that stays down for 20 ms (a long Hungarian sweep is < 1 ms) falls back to the copy.
static torch::Tensor read_counts(const Driver& d) {
    torch::Tensor counts;
    if (d.notify != nullptr) {
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (unsigned spins = 1;; ++spins) {
            if (__atomic_load_n(d.notify + 4, __ATOMIC_ACQUIRE) != 0) { seen = true; break; }
            if ((spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
            __builtin_ia32_pause();
        }
        if (seen) {
            counts = at::empty({4}, at::TensorOptions().dtype(torch::kInt32));
            std::memcpy(counts.data_ptr(), d.notify, 4 * sizeof(int32_t));
        }
    }
    if (!counts.defined()) counts = at::from_blob(d.small, {4}, d.iopts).cpu();
    return counts;
}

// Steady-state timesteps back to back (round 6: what the interpreter did between two timesteps -- unpacking the counts, the
// bookkeeping, building the next descriptor -- sat between the host read of one timestep and the first launch of the next, while
// the device waited).  steps: five integers per timestep (t, t_upto, next_t or -1, address of its det ids, D); state: N, A, E, Dn as
// the previous decode left them; limits: the one-launch kernels' row limit, the device solver's det limit (0: greedy) and whether
// the next timestep's FRONT may be enqueued before this timestep's counters are read (it then reads N and A on the device, the
// buffers sized by upper bounds: N <= the rows before the deletion, A <= the dets before it).  Stops BEFORE a timestep the native
// step does not take (no detections, the grown graph too large, a problem the device solver may not take: an early front of
// such a timestep is simply left behind -- it wrote beyond the rows in use, the caller's own update writes the same block again)
// and AFTER one whose status word is not clean or that ends the sequence; the caller goes on from there.  Returns {h', scores',
// counts of the last step} and {steps done, sum of E over their model calls, N, E, Dn, A, row-set flips, capacity of h' in rows}.
std::tuple<std::vector<torch::Tensor>, std::vector<int64_t>> greedy_run(std::vector<int64_t> ti, std::vector<int64_t> info, torch::Tensor h,
                                                                         int64_t cap_rows, std::vector<int64_t> steps,
                                                                         std::vector<int64_t> state, std::vector<int64_t> limits) {
    TORCH_CHECK(steps.size() % 5 == 0 && state.size() == 4 && limits.size() == 3, "greedy_run: bad descriptors");
    const Driver d = make_driver(ti, info, h);
    int64_t N = state[0], A = state[1], E = state[2], Dn = state[3];
    const int64_t max_rows = limits[0], hung_max = limits[1];
    const bool early = limits[2] != 0 && d.one_launch;
    TORCH_CHECK(h.defined() && h.dim() == 2 && h.size(0) == N && h.size(1) == d.GH && h.is_contiguous() &&
                    h.scalar_type() == torch::kFloat32, "greedy_run: h must be a contiguous fp32 [", N, ", ", d.GH, "] tensor");
    const tmpnn_track_rows* rows_cur = reinterpret_cast<const tmpnn_track_rows*>(ti[13]);
    const tmpnn_track_rows* rows_out = reinterpret_cast<const tmpnn_track_rows*>(ti[14]);
    int64_t done = 0, edges = 0, flips = 0;
    torch::Tensor sc, counts;
    Front front;
    const size_t K = steps.size() / 5;
    for (size_t k = 0; k < K; ++k) {
        const int64_t D = steps[5 * k + 4];
        if (D <= 0 || N == 0) break;
        const int64_t n_new = A * D + D, Nt = N + n_new;
        if (Nt > max_rows || (hung_max > 0 && Dn + D > hung_max)) break;
        const int t = (int)steps[5 * k], t_upto = (int)steps[5 * k + 1], next_t = (int)steps[5 * k + 2];
        // the next timestep's front, if it may go out before this one's counters are read: its bounds and the room it needs
        int64_t D2 = 0, N_ub = 0, A_ub = 0, n_ub = 0;
        bool early_next = false;
        if (early && next_t >= 0 && k + 1 < K) {
            D2 = steps[5 * (k + 1) + 4];
            N_ub = Nt; A_ub = Dn + D; n_ub = A_ub * D2 + D2;
            early_next = D2 > 0 && N_ub + n_ub <= max_rows && (hung_max == 0 || Dn + D + D2 <= hung_max);
        }
        const int64_t spare = std::max<int64_t>(2 * n_new + 256, early_next ? n_ub : 0);
        Back back;
        try {
            if (!front.valid)
                front = launch_front(d, N, A, D, t, reinterpret_cast<const int32_t*>(steps[5 * k + 3]), rows_cur, h, cap_rows, nullptr);
            back = launch_back(d, front, N, A, D, t_upto, next_t, rows_cur, rows_out, spare, early_next);
        } catch (const c10::Error&) {
            if (done == 0) throw;                  // (nothing changed yet: the caller sees the refusal itself)
            break;                                  // the caller's next call meets it as its first step
        }
        Front next_front;
        if (early_next)
            // (it also moves this timestep's kept state rows, which the decode above left in place: a refusal here -- its arguments
            //  are those of every other front -- leaves the state incomplete and is an error, not a fallback)
            next_front = launch_front(d, N_ub, A_ub, D2, (int)steps[5 * (k + 1)], reinterpret_cast<const int32_t*>(steps[5 * (k + 1) + 3]),
                                      rows_out, back.h_new, Nt + spare, d.small, back.h_out.data_ptr<float>());
        counts = read_counts(d);
        const int32_t* c = counts.data_ptr<int32_t>();
        edges += E + A * D;
        std::swap(rows_cur, rows_out);
        ++flips;
        cap_rows = Nt + spare;
        const int64_t n_keep = std::min<int64_t>(std::max<int64_t>(c[0], 0), Nt);
        h = back.h_new.narrow(0, 0, n_keep);
        sc = back.s_new.select(1, 0).narrow(0, 0, n_keep);
        N = n_keep; Dn = c[2]; E = N - Dn; A = c[3];
        front = next_front;
        ++done;
        if (hung_max > 0 && (c[1] & 2)) break;     // (a status the caller raises on)
        if (next_t < 0) break;
    }
    std::vector<torch::Tensor> out;
    if (done > 0) out = {h, sc, counts};
    return {out, {done, edges, N, E, Dn, A, flips, cap_rows}};
}

// ------------------------------------------------------------------------------------------------------------------------
// train.py:70-81 for one forward call as a native autograd node: tmpnn_train_losses_fwd / _bwd (one launch each way) without
// the interpreter's Function.apply / backward bookkeeping (~50 us per call of a batch-1 train chunk).  trackmpnn_amd/loss.py
// `_TrainLosses` is the same node in Python and stays the path for graphs the one-launch kernels do not take.
using tl_fwd_fn = int (*)(const tmpnn_graph*, const float*, const float*, const uint8_t*, int, uint8_t*, float*, float*, float*,
                          size_t, tmpnn_stream);
using tl_bwd_fn = int (*)(const tmpnn_graph*, const int32_t*, const int32_t*, const float*, const float*, const uint8_t*,
                          const float*, const float*, const float*, int, float*, float*, tmpnn_stream);

class TrainLossesFn : public torch::autograd::Function<TrainLossesFn> {
   public:
    // info: f_fwd, f_bwd, f_err, address of a struct tmpnn_graph (copied here), tp_classifier, src_pos ptr, dst_pos ptr, stream,
    //       workspace floats (tmpnn_train_losses_ws)
    static torch::autograd::variable_list forward(torch::autograd::AutogradContext* ctx, torch::Tensor logits, torch::Tensor scores,
                                                  torch::Tensor labels_u8, std::vector<int64_t> info,
                                                  std::vector<torch::Tensor> keep) {
        TORCH_CHECK(info.size() == 9, "train_losses: bad call descriptor");
        tmpnn_graph g;
        std::memcpy(&g, reinterpret_cast<const void*>(info[3]), sizeof(g));
        const int64_t Dn = g.Dn;
        auto opts = logits.options().dtype(torch::kFloat32).requires_grad(false);
        torch::Tensor lg = logits.detach().reshape({-1});
        if (lg.scalar_type() != torch::kFloat32 || !lg.is_contiguous()) lg = lg.to(torch::kFloat32).contiguous();
        torch::Tensor sc = scores.detach().reshape({-1});
        if (sc.scalar_type() != torch::kFloat32 || !sc.is_contiguous()) sc = sc.to(torch::kFloat32).contiguous();
        TORCH_CHECK(lg.numel() == g.N && sc.numel() == g.N && labels_u8.numel() == g.N && labels_u8.scalar_type() == torch::kUInt8 &&
                        labels_u8.is_contiguous(), "train_losses: logits / scores / labels must have one entry per row");
        torch::Tensor targets = at::empty_like(labels_u8);
        const int64_t nd8 = (Dn > 0 ? Dn : 1) * 8, n_ws = info[8];
        torch::Tensor buf = at::empty({nd8 + 4 + n_ws}, opts);
        float* base = buf.data_ptr<float>();
        const int rc = reinterpret_cast<tl_fwd_fn>(info[0])(&g, lg.data_ptr<float>(), sc.data_ptr<float>(),
                                                            labels_u8.data_ptr<uint8_t>(), (int)info[4], targets.data_ptr<uint8_t>(),
                                                            base, base + nd8, base + nd8 + 4, (size_t)n_ws,
                                                            reinterpret_cast<tmpnn_stream>(info[7]));
        TORCH_CHECK(rc == 0, "tmpnn_train_losses_fwd failed (code ", rc, "): ", reinterpret_cast<err_fn>(info[2])());
        ctx->saved_data["info"] = info;
        torch::Tensor gpod = at::empty({(int64_t)sizeof(tmpnn_graph)}, at::TensorOptions().dtype(torch::kByte));
        std::memcpy(gpod.data_ptr(), &g, sizeof(g));
        ctx->saved_data["g"] = gpod;
        ctx->saved_data["lg"] = lg;
        ctx->saved_data["sc"] = sc;
        ctx->saved_data["targets"] = targets;
        ctx->saved_data["buf"] = buf;
        ctx->saved_data["keep"] = keep;                     // (the graph's arena: the struct above points into it)
        ctx->saved_data["lshape"] = logits.sizes().vec();
        ctx->saved_data["sshape"] = scores.sizes().vec();
        ctx->set_materialize_grads(false);
        torch::Tensor out = buf.narrow(0, nd8, 4);
        return {out.select(0, 0), out.select(0, 3)};
    }

    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                   torch::autograd::variable_list grad_outputs) {
        std::vector<int64_t> info = ctx->saved_data["info"].toIntVector();
        torch::Tensor gpod = ctx->saved_data["g"].toTensor();
        const tmpnn_graph* g = reinterpret_cast<const tmpnn_graph*>(gpod.data_ptr());
        torch::Tensor lg = ctx->saved_data["lg"].toTensor(), sc = ctx->saved_data["sc"].toTensor();
        torch::Tensor targets = ctx->saved_data["targets"].toTensor(), buf = ctx->saved_data["buf"].toTensor();
        auto seed = [](const torch::Tensor& t) {
            torch::Tensor s = t.reshape({1});
            return (s.scalar_type() == torch::kFloat32 && s.is_contiguous()) ? s : s.to(torch::kFloat32).contiguous();
        };
        torch::Tensor d_c, d_f, d_logits, d_scores;
        if (grad_outputs[0].defined()) { d_c = seed(grad_outputs[0]); d_logits = at::empty_like(lg); }
        if (grad_outputs[1].defined()) { d_f = seed(grad_outputs[1]); d_scores = at::empty_like(sc); }
        const int rc = reinterpret_cast<tl_bwd_fn>(info[1])(
            g, reinterpret_cast<const int32_t*>(info[5]), reinterpret_cast<const int32_t*>(info[6]), lg.data_ptr<float>(),
            sc.data_ptr<float>(), targets.data_ptr<uint8_t>(), buf.data_ptr<float>(), d_c.defined() ? d_c.data_ptr<float>() : nullptr,
            d_f.defined() ? d_f.data_ptr<float>() : nullptr, (int)info[4], d_logits.defined() ? d_logits.data_ptr<float>() : nullptr,
            d_scores.defined() ? d_scores.data_ptr<float>() : nullptr, reinterpret_cast<tmpnn_stream>(info[7]));
        TORCH_CHECK(rc == 0, "tmpnn_train_losses_bwd failed (code ", rc, "): ", reinterpret_cast<err_fn>(info[2])());
        if (d_logits.defined()) d_logits = d_logits.reshape(ctx->saved_data["lshape"].toIntVector());
        if (d_scores.defined()) d_scores = d_scores.reshape(ctx->saved_data["sshape"].toIntVector());
        return {d_logits, d_scores, torch::Tensor(), torch::Tensor(), torch::Tensor()};
    }
};

std::vector<torch::Tensor> train_losses(torch::Tensor logits, torch::Tensor scores, torch::Tensor labels_u8, std::vector<int64_t> info,
                                        std::vector<torch::Tensor> keep) {
    return TrainLossesFn::apply(logits, scores, labels_u8, info, keep);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("train_losses", &train_losses, "create_targets + CELoss + the focal terms of one forward call as one native autograd node");
    m.def("small_iter", &small_iter, "fused TrackMPNN iteration on one small graph (in-place or sink gradient mode)");
    m.def("greedy_run", &greedy_run, "steady-state inference timesteps back to back (greedy_step in a loop, no interpreter between them)");
}
