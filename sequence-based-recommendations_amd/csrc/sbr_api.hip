// C-ABI of libsbr_rnn.so (see include/sbr_rnn.h): arena layout, Lasagne <-> device parameter
// layout conversion, and the orchestration of one training / prediction step.
// The host side mirrors what RNNBase does around its Theano functions
// (neural_networks/rnn_base.py:175-213, :285-300, :470-515); the arithmetic lives in the kernels.
#include "sbr_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <stdlib.h>
#include <math.h>
#include <atomic>
#include <chrono>

static thread_local std::string g_err;
void sbr_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    g_err = buf;
}
extern "C" const char* sbr_last_error(void) { return g_err.c_str(); }
extern "C" int sbr_abi_version(void) { return SBR_ABI_VERSION; }


// ---------------------------------------------------------------------------------------
// Layout
// ---------------------------------------------------------------------------------------
int sbr_build_layout(const sbr_config& cfg, Layout& lay, std::string& err) {
    char buf[256];
#define LFAIL(...) do { snprintf(buf, sizeof(buf), __VA_ARGS__); err = buf; return SBR_EINVAL; } while (0)
    if (cfg.abi_version != SBR_ABI_VERSION) LFAIL("abi_version %d != %d", cfg.abi_version, SBR_ABI_VERSION);
    if (cfg.cell < 0 || cfg.cell > 2) LFAIL("Unknown layer type %d", cfg.cell);                  // recurrent_layers.py:90
    if (cfg.loss < 0 || cfg.loss > SBR_LOSS_LIN) LFAIL("Unknown loss for the RNN model (%d)", cfg.loss);     // command_parser.py:123
    if (cfg.updater < 0 || cfg.updater > 4) LFAIL("Unknown update option %d", cfg.updater);       // update_manager.py:22
    if ((cfg.flags & SBR_FLAG_BF16_PROJECTION) && (cfg.flags & SBR_FLAG_F32_MFMA))
        LFAIL("SBR_FLAG_BF16_PROJECTION and SBR_FLAG_F32_MFMA contradict each other (bf16 inputs / exact f32 products for the output projection)");
    if (cfg.n_layers < 1 || cfg.n_layers > SBR_MAX_LAYERS) LFAIL("n_layers must be in [1,%d]", SBR_MAX_LAYERS);
    for (int l = 0; l < cfg.n_layers; ++l)
        if (cfg.layers[l] < 1 || cfg.layers[l] > 1024) LFAIL("layer %d size %d out of range [1,1024]", l, cfg.layers[l]);
    if (cfg.n_items < 1 || cfg.input_size < cfg.n_items) LFAIL("need n_items >= 1 and input_size >= n_items");
    if (cfg.n_feat < 1 || cfg.n_feat > 8) LFAIL("n_feat must be in [1,8]");
    if (cfg.max_length < 1) LFAIL("max_length must be >= 1");
    if (cfg.batch_size < 1 || cfg.local_batch < 1 || cfg.local_batch > cfg.batch_size) LFAIL("need 1 <= local_batch <= batch_size");
    if (cfg.row_offset < 0 || cfg.row_offset + cfg.local_batch > cfg.batch_size) LFAIL("row_offset/local_batch outside the global batch");
    const bool margin = SBR_LOSS_IS_MARGIN(cfg.loss);
    if (cfg.loss != SBR_LOSS_CCE && !margin && cfg.n_samples < 1) LFAIL("sampled losses need n_samples >= 1");
    if (margin && (cfg.n_targets < 1 || cfg.n_targets > 4096)) LFAIL("the multi-target losses need 1 <= n_targets <= 4096");
    if (cfg.learning_rate <= 0.0f) LFAIL("learning_rate must be > 0");
    if (cfg.embedding_size < 0 || cfg.embedding_size > 4096) LFAIL("embedding_size must be in [0,4096]");
#undef LFAIL
    lay = Layout();
    lay.cfg = cfg;
    lay.L = cfg.n_layers; lay.G = sbr_gates(cfg.cell); lay.T = cfg.max_length;
    lay.B = cfg.local_batch; lay.Bp = (cfg.local_batch + 15) / 16 * 16;
    lay.N = cfg.n_items; lay.F = cfg.n_feat; lay.Bg = cfg.batch_size;
    lay.S = (cfg.loss == SBR_LOSS_CCE || margin) ? 0 : cfg.n_samples;
    lay.C = lay.Bg + lay.S;
    lay.NT = margin ? cfg.n_targets : 1;
    const int G = lay.G, T = lay.T, Bp = lay.Bp;

    lay.E = cfg.embedding_size > 0 ? cfg.embedding_size : 0;
    lay.Ep = (lay.E + 3) / 4 * 4;
    lay.D = cfg.bidirectional ? 2 : 1;
    const int D = lay.D;
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += sbr_align(n); return o; };
    lay.p_Emb = lay.E ? take((size_t)cfg.input_size * lay.Ep) : 0;
    for (int l = 0; l < lay.L; ++l)
        for (int d = 0; d < D; ++d) {            // --r_bi: forward layer's parameters first (recurrent_layers.py:72-74)
            LayerLayout& y = lay.layer[l * D + d];
            y.H = cfg.layers[l]; y.Hp = sbr_pad_hidden(y.H); y.G = G;
            y.n_in = l == 0 ? (lay.E ? lay.F * lay.E : cfg.input_size) : D * cfg.layers[l - 1];
            y.n_in_p = l == 0 ? (lay.E ? lay.F * lay.Ep : cfg.input_size) : D * lay.layer[(l - 1) * D].Hp;
            y.p_Win = take((size_t)y.n_in_p * G * y.Hp);
            y.p_b = take((size_t)G * y.Hp);
            y.p_Whid = take((size_t)y.Hp * G * y.Hp);
            y.p_peep = take((size_t)3 * y.Hp);
            y.p_cinit = take(y.Hp);
            y.p_hinit = take(y.Hp);
        }
    lay.HLp = lay.layer[(lay.L - 1) * D].Hp;
    lay.HLt = D * lay.HLp;
    lay.p_split = off;
    lay.p_WoutT = take((size_t)lay.N * lay.HLt);
    lay.p_bout = take(lay.N);
    lay.n_params = off;
    lay.n_state_arrays = (cfg.updater == SBR_UPD_ADAM || cfg.updater == SBR_UPD_ADADELTA) ? 2 : 1;

    lay.s_params = 0;
    lay.s_grads = sbr_align(lay.n_params);
    lay.s_state = lay.s_grads + sbr_align(lay.n_params + 1);
    lay.s_act = lay.s_state + lay.n_state_arrays * sbr_align(lay.n_params);
    // NB: state arrays are addressed as s_state + k*n_params; n_params is already 64-aligned.

    off = 0;
    size_t maxrec = 0, max_dense_in = 0;
    for (int pl = 0; pl < lay.L * D; ++pl) {
        LayerLayout& y = lay.layer[pl];
        const int l = pl / D;
        const size_t tb = (size_t)T * Bp, tb1 = (size_t)(T + 1) * Bp;
        y.a_xt = take(tb * G * y.Hp);
        y.a_hs = take(tb1 * y.Hp);
        y.a_cs = cfg.cell == SBR_CELL_LSTM ? take(tb1 * y.Hp) : 0;
        const bool wide = y.Hp == 256 || y.Hp == 512;
        y.a_xh = wide ? take((size_t)4 * Bp * y.Hp) : 0;
        y.a_pring = wide ? take(sbr_rec_c16_ring_floats(Bp, y.Hp)) : 0;
        for (int k = 0; k < 4; ++k) y.a_g[k] = cfg.cell == SBR_CELL_VANILLA ? 0 : take(tb * y.Hp);
        y.a_dxt = take(tb * G * y.Hp);
        y.a_dhi = cfg.cell == SBR_CELL_GRU ? take(tb * y.Hp) : y.a_dxt;
        y.a_dhext = l < lay.L - 1 ? take(tb * y.Hp) : 0;
        y.a_state = take((size_t)2 * Bp * y.Hp);
        y.a_part = take((size_t)SBR_BWD_CHUNKS * Bp * (G * y.Hp + 5 * y.Hp));
        maxrec = std::max(maxrec, (size_t)y.Hp * G * y.Hp);
        if (l > 0 || lay.E) { maxrec = std::max(maxrec, (size_t)y.n_in_p * G * y.Hp); max_dense_in = std::max(max_dense_in, (size_t)y.n_in_p); }
    }
    if (lay.E) { lay.a_emb = take((size_t)T * Bp * lay.F * lay.Ep); lay.a_demb = take((size_t)T * Bp * lay.F * lay.Ep); }
    if (D == 2) {
        lay.a_Xr = take((size_t)Bp * T * lay.F);
        if (lay.E) lay.a_embr = take((size_t)T * Bp * lay.F * lay.Ep);
        for (int l = 0; l + 1 < lay.L; ++l) {
            lay.a_cat[l] = take((size_t)T * Bp * 2 * lay.layer[l * 2].Hp);
            lay.a_catr[l] = take((size_t)T * Bp * 2 * lay.layer[l * 2].Hp);
        }
        lay.a_hcat = take((size_t)Bp * lay.HLt);
        lay.a_dhl[0] = take((size_t)Bp * lay.HLp); lay.a_dhl[1] = take((size_t)Bp * lay.HLp);
        if (max_dense_in) { lay.a_dinp[0] = take((size_t)T * Bp * max_dense_in); lay.a_dinp[1] = take((size_t)T * Bp * max_dense_in); }
        if (!lay.E) {
            lay.a_s2cnt = take((size_t)cfg.input_size + 1); lay.a_s2off = take((size_t)cfg.input_size + 1);
            lay.a_s2cur = take((size_t)cfg.input_size + 1);
            lay.a_s2sid = take((size_t)T * Bp * lay.F); lay.a_s2pos = take((size_t)T * Bp * lay.F);
        }
    }
    lay.a_logits = take((size_t)Bp * ((lay.N + 3) & ~3));
    lay.a_dhlast = take((size_t)Bp * lay.HLt);
    lay.a_rowcost = take(Bp);
    if (lay.S > 0) {
        lay.a_Wc = take((size_t)lay.C * lay.HLt); lay.a_bc = take(lay.C);
        lay.a_act = take((size_t)Bp * lay.C);
        lay.a_dWc = take((size_t)lay.C * lay.HLt); lay.a_dbc = take(lay.C);
    }
    lay.a_csum = take((size_t)16 * std::max(lay.N, lay.C));
    lay.a_prof = take((size_t)3 * (Bp / 16) * 16 * 8 * 2);      // forward | backward | the one-launch head (tools/rec_prof.py, cl_prof.py, head_prof.py)
    lay.a_fault = take(64);
    lay.a_clx = take((size_t)Bp * 8 + 64);      // handshake slots: up to Bp/4 tiles x 32 members
    lay.ws_floats = std::max((size_t)1 << 20, 64 * maxrec);
    lay.a_ws = take(lay.ws_floats);
    lay.ws2_floats = 4 * lay.ws_floats;          // up to 256 weight-gradient slabs
    lay.a_ws2 = take(lay.ws2_floats);
    // the split-K workspace of the dW_out GEMM when SBR_TAIL_OUT_STREAM moves it off the side stream (the polling weight-gradient
    // GEMM owns ws2 meanwhile): an experiment switch -- taken from the arena only when it is set
    lay.ws3_floats = 0;      // (the output layer's dW_out GEMM on a stream of its own: measured slower, profiles/round4_variants.txt calls n, q)
    lay.a_ws3 = lay.ws3_floats ? take(lay.ws3_floats) : 0;
    lay.a_X = take((size_t)Bp * T * lay.F);
    lay.a_len = take(Bp);
    lay.a_tgt = take((size_t)std::max(lay.Bg, Bp) * lay.NT);
    lay.a_dflt = margin ? take(lay.N) : 0;
    lay.a_smp = take(std::max(lay.S, 1));
    lay.a_cells = take(std::max(lay.C, 1));
    lay.a_pop = take(Bp);
    lay.a_topk = take((size_t)Bp * 64);
    lay.a_X2 = take((size_t)Bp * T * lay.F);
    lay.a_len2 = take(Bp);
    lay.a_tgt2 = take((size_t)std::max(lay.Bg, Bp) * lay.NT);
    lay.a_smp2 = take(std::max(lay.S, 1));
    lay.a_pop2 = take(Bp);
    // tail overlap (sbr_backward_recurrent): the sort's keys carry a time chunk, its counters cover chunks x ids
    lay.tail_keys = 1;
    if (lay.L == 1 && D == 1 && !lay.E && T >= 64 && T < 4096)
        lay.tail_keys = std::max(1, std::min(8, sbr_scatter_lds_ids() / std::max(1, cfg.input_size)));
    lay.a_scnt = take((size_t)lay.tail_keys * cfg.input_size + 1); lay.a_soff = take((size_t)lay.tail_keys * cfg.input_size + 1);
    lay.a_scur = take((size_t)lay.tail_keys * cfg.input_size + 1);
    lay.a_sP = take((size_t)cfg.input_size + 2);
    lay.a_sid = take((size_t)T * Bp * lay.F); lay.a_spos = take((size_t)T * Bp * lay.F);
    {   // wide index-input rows: the partial rows of the long segments' pieces (launch_scatter_wide)
        const int ghp0 = G * lay.layer[0].Hp;
        const bool wide0 = !lay.E && ghp0 >= 512 && ghp0 <= 8192;
        lay.sr_slots = wide0 ? std::max(sbr_scatter_wide_slots((size_t)T * Bp * lay.F), 2 * SBR_SCAT_RANGES) : 0;
        lay.a_srpart = wide0 ? take((size_t)lay.sr_slots * ghp0) : 0;
        lay.a_srid = wide0 ? take((size_t)lay.sr_slots * 4 + 8) : 0;
    }
    lay.a_hstat = take((size_t)256 * 64 + 64);      // + the head's arrival counter and done flag
    lay.a_prog = take((size_t)Bp * 2 + 256);      // per-wave words, (a gap), the chain's clock words
    lay.a_done = take((size_t)SBR_DONE_COPIES * SBR_DONE_STRIDE);      // the monitor's word, replicated (sbr_common.h SbrPoll)
    // Row-sparse blocks (sbr_sparse.hip): the index-addressed rows of layer 0 (or of the embedding table) and, for the sampled
    // heads, the rows of W_out^T / b_out.  Taken when a step cannot touch every row anyway (more rows than candidates) or
    // when the flag forces it; SBR_FLAG_DENSE_UPDATE keeps the dense Lasagne-style pass over everything.
    lay.n_sparse = 0; lay.a_at = 0; lay.n_at = 0; lay.adam_early_exit = 0;
    // Adam's lazy replay walks a row's missed steps one by one and stops when the momentum term can no longer move the row -- at
    // most SBR_ADAM_REPLAY_CAP steps (sbr_sparse.hip).  With beta1 so close to 1 that the momentum is still alive there
    // (beta1^cap > 1e-9: beta1 > 0.9975) rows untouched for longer would lose the rest of their updates: such configurations keep
    // the dense Lasagne-style pass (exact, only slower); forcing the sparse form for them is refused.
    const bool adam_replay_unbounded = cfg.updater == SBR_UPD_ADAM && pow((double)cfg.beta1, 8190.0) > 1e-9;
    if (adam_replay_unbounded && (cfg.flags & SBR_FLAG_SPARSE_UPDATE)) {
        snprintf(buf, sizeof(buf), "row-sparse Adam steps need beta1 <= 0.9975 (beta1 = %g keeps a row's momentum alive beyond the replay cap)",
                 (double)cfg.beta1);
        err = buf; return SBR_EINVAL;
    }
    if (!(cfg.flags & SBR_FLAG_DENSE_UPDATE) && !adam_replay_unbounded) {
        const bool force = cfg.flags & SBR_FLAG_SPARSE_UPDATE;
        const int world = (lay.Bg + lay.B - 1) / lay.B;
        const long cand0 = (long)T * Bp * lay.F;
        if (force || (long)cfg.input_size > cand0) {
            SparseBlockLayout& b = lay.sparse[lay.n_sparse++];
            b = SparseBlockLayout(); b.kind = 0; b.n_rows = cfg.input_size;
            if (lay.E) { b.npairs = 1; b.off[0] = lay.p_Emb; b.width[0] = lay.Ep; b.stride[0] = lay.Ep; }
            else {
                b.npairs = D;
                for (int d = 0; d < D; ++d) { b.off[d] = lay.layer[d].p_Win; b.width[d] = G * lay.layer[d].Hp; b.stride[d] = G * lay.layer[d].Hp; }
            }
            b.max_local = (int)std::min<long>(b.n_rows, cand0);
        }
        if (lay.S > 0 && (force || lay.N > lay.C)) {
            SparseBlockLayout& b = lay.sparse[lay.n_sparse++];
            b = SparseBlockLayout(); b.kind = 1; b.n_rows = lay.N; b.npairs = 2;
            b.off[0] = lay.p_WoutT; b.width[0] = lay.HLt; b.stride[0] = lay.HLt;
            b.off[1] = lay.p_bout; b.width[1] = 1; b.stride[1] = 1;
            b.max_local = std::min(lay.N, lay.C);
        }
        for (int i = 0; i < lay.n_sparse; ++i) {
            SparseBlockLayout& b = lay.sparse[i];
            b.W = 0; for (int k = 0; k < b.npairs; ++k) b.W += b.width[k];
            b.cand_cap = world * b.max_local + 64;
            b.a_last = take(b.n_rows); b.a_mark = take(b.n_rows); b.a_cand = take(b.cand_cap); b.a_count = take(64);
        }
        if (lay.n_sparse && cfg.updater == SBR_UPD_ADAM) {
            // a_t = lr sqrt(1 - b2^t) / (1 - b1^t) as the dense launch computes it (double, rounded to float), tabulated until
            // it has been == (float)lr for a while; the catch-up replays missed steps with their own a_t
            const double lr = cfg.learning_rate, b1 = cfg.beta1, b2 = cfg.beta2;
            int t = 1, same = 0; double worst = 0.0, prev = 0.0;
            const int cap = 1 << 22;
            for (; t <= cap && same < 256; ++t) {
                const double a = lr * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t));
                same = ((float)a == (float)lr) ? same + 1 : 0;
                if (t > 1) worst = std::max(worst, a / prev);
                prev = a;
            }
            lay.n_at = t - 1;
            lay.adam_early_exit = (b2 > 0.0 && worst * b1 / sqrt(b2) < 0.999) ? 1 : 0;
            lay.a_at = take(lay.n_at);
        }
    }
    lay.s_end = lay.s_act + off;
    return SBR_OK;
}

void sbr_param_descs(const Layout& lay, std::vector<ParamDesc>& out) {
    out.clear();
    const int cell = lay.cfg.cell;
    static const char* lstm_g[4] = {"ingate", "forgetgate", "cell", "outgate"};       // sparse_lstm.py:240-254
    static const char* gru_g[3] = {"updategate", "resetgate", "hidden_update"};       // sparse_lstm.py:660-668
    static const char* van_g[1] = {"hidden_update"};
    const char* const* gn = cell == SBR_CELL_LSTM ? lstm_g : (cell == SBR_CELL_GRU ? gru_g : van_g);
    if (lay.E) out.push_back({"emb.W", 0, 8, 0, lay.cfg.input_size, lay.E, 2});   // lasagne EmbeddingLayer comes first
    for (int pl = 0; pl < lay.L * lay.D; ++pl) {
        const LayerLayout& y = lay.layer[pl];
        const int l = pl;      // ParamDesc.layer indexes lay.layer[] (level * D + direction)
        char pre[16];
        if (lay.D == 1) snprintf(pre, sizeof(pre), "l%d.", pl);
        else snprintf(pre, sizeof(pre), "l%d%c.", pl / 2, "fb"[pl & 1]);
        if (cell == SBR_CELL_VANILLA && (pl / lay.D > 0 || lay.E)) {
            // dense input: stock lasagne RecurrentLayer = CustomRecurrentLayer over two DenseLayers, whose get_params lists
            // its own parameter first and then the children's (recurrent_layers.py:94-104 [3P])
            out.push_back({std::string(pre) + "hid_init", l, 5, 0, 1, y.H, 2});
            out.push_back({std::string(pre) + "input_to_hidden.W", l, 0, 0, y.n_in, y.H, 2});
            out.push_back({std::string(pre) + "input_to_hidden.b", l, 2, 0, y.H, 1, 1});
            out.push_back({std::string(pre) + "hidden_to_hidden.W", l, 1, 0, y.H, y.H, 2});
            continue;
        }
        for (int g = 0; g < lay.G; ++g) {
            out.push_back({std::string(pre) + "W_in_to_" + gn[g], l, 0, g, y.n_in, y.H, 2});
            out.push_back({std::string(pre) + "W_hid_to_" + gn[g], l, 1, g, y.H, y.H, 2});
            out.push_back({std::string(pre) + "b_" + gn[g], l, 2, g, y.H, 1, 1});
        }
        if (cell == SBR_CELL_LSTM) {
            out.push_back({std::string(pre) + "W_cell_to_ingate", l, 3, 0, y.H, 1, 1});
            out.push_back({std::string(pre) + "W_cell_to_forgetgate", l, 3, 1, y.H, 1, 1});
            out.push_back({std::string(pre) + "W_cell_to_outgate", l, 3, 2, y.H, 1, 1});
            out.push_back({std::string(pre) + "cell_init", l, 4, 0, 1, y.H, 2});
        }
        out.push_back({std::string(pre) + "hid_init", l, 5, 0, 1, y.H, 2});
    }
    out.push_back({"out.W", (lay.L - 1) * lay.D, 6, 0, lay.D * lay.layer[(lay.L - 1) * lay.D].H, lay.N, 2});
    out.push_back({"out.b", (lay.L - 1) * lay.D, 7, 0, lay.N, 1, 1});
}

// position of Lasagne gate g (creation order) inside the stacked matrices:
// LSTM [i,f,c,o] = creation order (sparse_lstm.py:348-360); GRU stacks [reset, update, hidden]
// although it creates update first (sparse_lstm.py:737-749).
static inline int stacked_pos(int cell, int g) { return cell == SBR_CELL_GRU ? (g == 0 ? 1 : (g == 1 ? 0 : 2)) : g; }

// copy one Lasagne array <-> its place in a host image of a parameter-shaped section
static void convert_param(const Layout& lay, const ParamDesc& d, float* image, float* arr, bool to_image) {
    const LayerLayout& y = lay.layer[d.layer];
    const int GHp = y.G * y.Hp;
    auto mv = [&](size_t io, size_t ao) { if (to_image) image[io] = arr[ao]; else arr[ao] = image[io]; };
    const int gp = stacked_pos(lay.cfg.cell, d.gate);
    switch (d.kind) {
        case 0:
            for (int64_t r = 0; r < d.d0; ++r) {
                // stored row of logical input row r: behind an embedding f*E + e -> f*Ep + e; above another level its
                // D concatenated outputs of H units each are stored Hp apart
                int64_t rs = r;
                const int level = d.layer / lay.D;
                if (level == 0 && lay.E) rs = (r / lay.E) * lay.Ep + r % lay.E;
                else if (level > 0) { const LayerLayout& lo = lay.layer[(level - 1) * lay.D]; rs = (r / lo.H) * lo.Hp + r % lo.H; }
                for (int64_t c = 0; c < d.d1; ++c) mv(y.p_Win + rs * GHp + gp * y.Hp + c, r * d.d1 + c);
            }
            break;
        case 8: for (int64_t r = 0; r < d.d0; ++r) for (int64_t c = 0; c < d.d1; ++c) mv(lay.p_Emb + r * lay.Ep + c, r * d.d1 + c); break;
        case 1: for (int64_t r = 0; r < d.d0; ++r) for (int64_t c = 0; c < d.d1; ++c) mv(y.p_Whid + r * GHp + gp * y.Hp + c, r * d.d1 + c); break;
        case 2: for (int64_t c = 0; c < d.d0; ++c) mv(y.p_b + gp * y.Hp + c, c); break;
        case 3: for (int64_t c = 0; c < d.d0; ++c) mv(y.p_peep + d.gate * y.Hp + c, c); break;
        case 4: for (int64_t c = 0; c < d.d1; ++c) mv(y.p_cinit + c, c); break;
        case 5: for (int64_t c = 0; c < d.d1; ++c) mv(y.p_hinit + c, c); break;
        case 6:
            for (int64_t k = 0; k < d.d0; ++k) {
                const int64_t ks = (k / y.H) * lay.HLp + k % y.H;     // --r_bi: [forward H | backwards H] stored HLp apart
                for (int64_t n = 0; n < d.d1; ++n) mv(lay.p_WoutT + n * lay.HLt + ks, k * d.d1 + n);
            }
            break;
        case 7: for (int64_t n = 0; n < d.d0; ++n) mv(lay.p_bout + n, n); break;
    }
}

// ---------------------------------------------------------------------------------------
// create / destroy / parameters
// ---------------------------------------------------------------------------------------
extern "C" int sbr_arena_bytes(const sbr_config* cfg, size_t* bytes) {
    CHECK_ARG(cfg && bytes, "null argument");
    Layout lay; std::string err;
    if (sbr_build_layout(*cfg, lay, err) != SBR_OK) { sbr_set_error("%s", err.c_str()); return SBR_EINVAL; }
    *bytes = lay.s_end * sizeof(float);
    return SBR_OK;
}

extern "C" int sbr_create(const sbr_config* cfg, void* arena, size_t arena_bytes, void* stream, sbr_handle** out) {
    CHECK_ARG(cfg && out, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        sbr_set_error("no HIP device visible: libsbr_rnn.so has no CPU path");
        return SBR_EHIP;
    }
    sbr_handle* h = new sbr_handle();
    std::string err;
    if (sbr_build_layout(*cfg, h->lay, err) != SBR_OK) { sbr_set_error("%s", err.c_str()); delete h; return SBR_EINVAL; }
    const size_t need = h->lay.s_end * sizeof(float);
    h->stream = (hipStream_t)stream;
    if (arena) {
        if (arena_bytes < need || ((uintptr_t)arena & 255)) {
            sbr_set_error("arena too small (%zu < %zu bytes) or not 256-byte aligned", arena_bytes, need);
            delete h; return SBR_EINVAL;
        }
        h->arena = (float*)arena; h->own_arena = false;
    } else {
        void* p = nullptr;
        if (hipMalloc(&p, need) != hipSuccess) { sbr_set_error("hipMalloc(%zu) failed", need); delete h; return SBR_ENOMEM; }
        h->arena = (float*)p; h->own_arena = true;
    }
    sbr_param_descs(h->lay, h->descs);
    h->rpt = 16;
    {   // rows per workgroup of the bf16x6 recurrent kernels: the per-step latency of the chain does not depend on the
        // tile height, so take the smallest tile whose workgroups still fit the 256 CUs in one round (measured, GRU-128
        // T=200: B=512 548k seq/s at 4 rows vs 444k at 8; B=1024 856k vs 766k at 8 / 643k at 16; B=2048 1078k at 8 vs
        // 922k at 4 (two rounds) / 1047k at 16; B=4096 1380k at 16 vs 1124k at 8)
        const char* e = getenv("SBR_RPT");
        int r = e ? atoi(e) : 0;
        if (r != 1 && r != 2 && r != 4 && r != 8 && r != 16) { r = 4; while (r < 16 && h->lay.Bp / r > 256) r <<= 1; }
        h->rpt = r;
    }
    {
        const char* e = getenv("SBR_BWD_CHUNKS");
        const int c = e ? atoi(e) : 1;   // chunking measured neutral-to-slower at C2 (relaunch ~20 us); kept + tested
        h->bwd_chunks = c < 1 ? 1 : (c > SBR_BWD_CHUNKS ? SBR_BWD_CHUNKS : c);
        h->wgrad_slices = 256;
        const char* cl = getenv("SBR_CLUSTER");
        h->cluster = cl ? atoi(cl) != 0 : 1;
        const char* ln = getenv("SBR_CL_LINEAR");
        h->cl_linear = ln ? atoi(ln) != 0 : 0;
        h->cl_epoch = 0;
        h->wgrad_x6 = 1;
        h->x6_split = 1;
        const char* xp = getenv("SBR_X6_PIPE");
        h->x6_pipe = xp ? atoi(xp) : 1;   // 0: barrier kernels (x6s), 1: pipelined without the matrix-pipe gate, 2: with it (rounds 1-3: with
                                          // one sparse instruction per k-block the partner's phase is over long before, the gate only costs its read)
        const char* fg = getenv("SBR_FUSE_GATHER");
        h->fuse_gather = fg ? atoi(fg) != 0 : 1;
    }
    h->n_rows = 0; h->step_count = 0; h->have_batch = false; h->fwd_done = false; h->timing = false;
    h->grads_clean = false; h->timing_marks = 0; h->marks_shared = 0; h->tail_swapped = false;
    h->swap_tail = true;
    { const char* e = getenv("SBR_TAIL_OVERLAP"); h->tail_overlap = e ? atoi(e) : 1; }
    // Tuned constants of the overlapped tail (each was an environment switch while it was being measured -- rounds 2 - 5; the A/B
    // numbers are in profiles/round2_b_tail_variants.txt, round3_*_variants.txt, round5_variants.txt and DESIGN.md sections 3, 3a, 3d)
    h->tail_chunks_max = 8; h->tail_pub_every = 2; h->tail_short_chunks = 3;
    // (every switch is read here, once per handle: a test that flips one between two engines of a process gets what it asked for)
    if (getenv("SBR_TAIL_TRACE") && atoi(getenv("SBR_TAIL_TRACE"))) {      // tools/tail_trace.py
        if (hipMalloc(&h->tail_trace, 16384 * sizeof(unsigned long long)) != hipSuccess) h->tail_trace = nullptr;
        else (void)hipMemset(h->tail_trace, 0, 16384 * sizeof(unsigned long long));
    }
    h->tail_fence_kb = 124; h->tail_early_sort = 1; h->tail_out_stream = 0; h->tail_fuse_slabs = 1; h->tail_slab_growth = 0.35;
    { const char* e = getenv("SBR_TAIL_SCATTER_LDS"); h->tail_scatter_lds = e ? atoi(e) : 1; }      // 0: the polling range form (also the way out when the LDS rows run out)
    h->tail_geom = h->tail_scatter_lds ? 1.6 : 2.6;
    h->tail_mon_units = 1; h->tail_first = 6; h->tail_scatter_units = 192; h->tail_gemm_groups = 64; h->tail_slab_max = 512;
    h->fold_dh = true;
    { const char* e = getenv("SBR_WGRAD_F16"); h->wgrad_f16 = e ? atoi(e) : 1; }
    h->wgrad_x6_wgs = 512;
    h->tail_nc = 0; h->tail_ch = 0; h->prog_epoch = 0; h->tail_updated = false; h->ev_tail = nullptr; h->ev_tail2 = nullptr; h->side2 = nullptr; h->ev_lg_rec = nullptr; h->side3 = nullptr; h->ev_tail3 = nullptr; h->tail_sorted = false; h->out3 = false;
    h->step_open = false; h->tail_join_pending = false;
    memset(h->ev, 0, sizeof(h->ev)); h->ring_used = 0; h->ring_cur = 0;
    memset(h->ev_ch, 0, sizeof(h->ev_ch)); h->ch_n = 0; h->chain_timing = false;
    h->side = nullptr; h->ev_fork = nullptr; h->ev_join = nullptr; h->ev_sort = nullptr; h->ev_lg = nullptr; h->ev_fill = nullptr; h->ev_og = nullptr;
    for (int c = 0; c < SBR_BWD_CHUNKS; ++c) h->ev_chunk[c] = nullptr;
    h->in_train_step = false; h->side_pending = false; h->deferred_join = false; h->fill_done = false; h->og_recorded = false;
    h->out_early = false; h->dh_slabs_n = 0;
    h->s_bb = nullptr; h->ev_bb = nullptr; h->ev_bbw = nullptr; h->bb_set = 0; h->batch_seq = 0; h->set_use[0] = h->set_use[1] = 0;
    h->lg_seq = 0; h->train_fwd_open = false; h->bb_slow = 0; h->bb_unread = false;
    { const char* e = getenv("SBR_SPARSE_OUT_EARLY"); h->sparse_out_early = e ? atoi(e) : 1; }
    h->cells_early = false; h->wout_early = false;
    h->ev_cells = nullptr;
    { const char* e = getenv("SBR_HEAD_FUSE"); h->head_fuse = e ? atoi(e) : 1; }
    { const char* e = getenv("SBR_OUT_FUSE"); h->out_fuse = e ? atoi(e) != 0 : 1; }
    { const char* e = getenv("SBR_ROW_AWARE_UPDATE"); h->row_aware = e ? atoi(e) != 0 : 1; }
    h->out_stepped = false;
    h->head_epoch = 0;
    h->lag_host = nullptr; h->lag_slot = 0; h->lag_pending = -1; h->lag_counter = 0; h->lag_seq[0] = h->lag_seq[1] = 0;
    // The side stream must not share a hardware queue with the main stream (HIP multiplexes streams onto
    // GPU_MAX_HW_QUEUES = 4 queues; with RCCL's streams alive the side stream landed on the main stream's queue and
    // every "overlapped" kernel serialised: +150 us per step in the data-parallel path).  Streams of another priority
    // level get their own queues.
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_sort, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_lg, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fill, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_og, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_tail2, hipEventDisableTiming) != hipSuccess ||
        hipStreamCreateWithPriority(&h->side2, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&h->side3, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_tail3, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_cells, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_chunk[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_chunk[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_chunk[2], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_chunk[3], hipEventDisableTiming) != hipSuccess ||
#if SBR_BB_STREAM
        hipStreamCreateWithPriority(&h->s_bb, hipStreamNonBlocking, SBR_BB_STREAM == 2 ? prio_lo : prio_hi) != hipSuccess ||
#endif
        hipEventCreateWithFlags(&h->ev_bb, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_bbw, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void**)&h->lag_host, 8 * sizeof(float), hipHostMallocDefault) != hipSuccess) {
        sbr_set_error("side stream creation failed"); sbr_destroy(h); return SBR_EHIP;
    }
#if !SBR_BB_STREAM
    h->s_bb = h->side3;      // (NOT a fifth stream: the device serves four hardware queues per process, a fifth stream shares one with the
                             //  main stream and the step takes 0.76 ms instead of 0.33 -- profiles/round6_variants.txt, call m)
#endif
    memset(h->lag_host, 0, 8 * sizeof(float));
    // parameters, gradients, optimizer state and batch buffers start as zeros
    // (activations too: one-off, keeps every later GEMM operand finite)
    hipError_t e = hipMemsetAsync(h->arena, 0, h->lay.s_end * sizeof(float), h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { sbr_set_error("arena initialisation failed: %s", hipGetErrorString(e)); sbr_destroy(h); return SBR_EHIP; }
    h->sp_exchanged[0] = h->sp_exchanged[1] = 0; h->sp_ncand[0] = h->sp_ncand[1] = 0; h->sp_epoch = 0;
    if (h->lay.n_at > 0) {
        std::vector<float> at(h->lay.n_at);
        const double lr = cfg->learning_rate, b1 = cfg->beta1, b2 = cfg->beta2;
        for (int t = 1; t <= h->lay.n_at; ++t) at[t - 1] = (float)(lr * sqrt(1.0 - pow(b2, (double)t)) / (1.0 - pow(b1, (double)t)));
        e = hipMemcpy(h->A(h->lay.a_at), at.data(), at.size() * sizeof(float), hipMemcpyHostToDevice);
        if (e != hipSuccess) { sbr_set_error("a_t table upload failed: %s", hipGetErrorString(e)); sbr_destroy(h); return SBR_EHIP; }
    }
    *out = h;
    return SBR_OK;
}

extern "C" void sbr_destroy(sbr_handle* h) {
    if (!h) return;
    // work still in flight may write into what is freed below (the lagged step's report into pinned memory, a batch build into the
    // arena's second set): let every stream of the engine drain first
    (void)hipStreamSynchronize(h->stream);
    if (h->side) (void)hipStreamSynchronize(h->side);
    if (h->side2) (void)hipStreamSynchronize(h->side2);
    if (h->side3) (void)hipStreamSynchronize(h->side3);
    for (int r = 0; r < sbr_handle::kRing; ++r)
        for (int i = 0; i < SBR_N_PHASES; ++i) if (h->ev[r][i]) (void)hipEventDestroy(h->ev[r][i]);
    for (int r = 0; r < sbr_handle::kChain; ++r)
        for (int i = 0; i < 2; ++i) if (h->ev_ch[r][i]) (void)hipEventDestroy(h->ev_ch[r][i]);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_sort) (void)hipEventDestroy(h->ev_sort);
    if (h->ev_lg) (void)hipEventDestroy(h->ev_lg);
    if (h->ev_fill) (void)hipEventDestroy(h->ev_fill);
    if (h->ev_og) (void)hipEventDestroy(h->ev_og);
    if (h->ev_tail) (void)hipEventDestroy(h->ev_tail);
    if (h->ev_tail2) (void)hipEventDestroy(h->ev_tail2);
    if (h->side2) (void)hipStreamDestroy(h->side2);
#if SBR_BB_STREAM
    if (h->s_bb) (void)hipStreamDestroy(h->s_bb);
#endif
    if (h->ev_bb) (void)hipEventDestroy(h->ev_bb);
    if (h->ev_bbw) (void)hipEventDestroy(h->ev_bbw);
    if (h->side3) (void)hipStreamDestroy(h->side3);
    if (h->ev_tail3) (void)hipEventDestroy(h->ev_tail3);
    if (h->ev_cells) (void)hipEventDestroy(h->ev_cells);
    for (int c = 0; c < SBR_BWD_CHUNKS; ++c) if (h->ev_chunk[c]) (void)hipEventDestroy(h->ev_chunk[c]);
    if (h->lag_host) (void)hipHostFree(h->lag_host);
    if (h->own_arena && h->arena) (void)hipFree(h->arena);
    if (h->tail_trace) (void)hipFree(h->tail_trace);
    if (h->tail_slab_dev) (void)hipFree(h->tail_slab_dev);
    delete h;
}

extern "C" int sbr_num_params(const sbr_handle* h) { return h ? (int)h->descs.size() : SBR_EINVAL; }

extern "C" int sbr_param_shape(const sbr_handle* h, int i, int64_t dims[2], int* ndim) {
    CHECK_ARG(h && dims && ndim && i >= 0 && i < (int)h->descs.size(), "bad parameter index %d", i);
    const ParamDesc& d = h->descs[i];
    dims[0] = d.d0; dims[1] = d.ndim == 2 ? d.d1 : 1; *ndim = d.ndim;
    return SBR_OK;
}

// ---------------------------------------------------------------------------------------
// row-sparse blocks (sbr_sparse.hip)
// ---------------------------------------------------------------------------------------
static SbrSparseRows sparse_rows(sbr_handle* h, int b) {
    const SparseBlockLayout& sb = h->lay.sparse[b];
    SbrSparseRows r; memset(&r, 0, sizeof(r));
    r.npairs = sb.npairs; r.n_rows = sb.n_rows;
    for (int k = 0; k < sb.npairs; ++k) { r.off[k] = sb.off[k]; r.width[k] = sb.width[k]; r.stride[k] = sb.stride[k]; }
    r.p = h->P(0); r.g = h->Gd(0); r.s0 = h->St(0, 0); r.s1 = h->lay.n_state_arrays > 1 ? h->St(1, 0) : nullptr;
    r.last = (int*)h->A(sb.a_last);
    return r;
}
static SbrSparseUpd sparse_upd(sbr_handle* h) {
    const sbr_config& c = h->lay.cfg;
    SbrSparseUpd u; u.updater = c.updater; u.lr = c.learning_rate; u.rho = c.rho; u.b1 = c.beta1; u.b2 = c.beta2;
    u.at = h->lay.n_at ? h->A(h->lay.a_at) : nullptr; u.n_at = h->lay.n_at; u.early_exit = h->lay.adam_early_exit;
    return u;
}
// adagrad's zero-gradient step is a no-op: nothing is ever pending
static inline bool sparse_lazy(const sbr_handle* h) { return h->lay.n_sparse > 0 && h->lay.cfg.updater != SBR_UPD_ADAGRAD; }

// every row of every sparse block current through the last applied step (before parameters are read as a whole)
static int flush_lazy(sbr_handle* h, int only_kind = -1) {
    if (!sparse_lazy(h)) return SBR_OK;
    for (int b = 0; b < h->lay.n_sparse; ++b)
        if (only_kind < 0 || h->lay.sparse[b].kind == only_kind)
            SBR_LAUNCH(launch_sparse_flush(h->stream, sparse_rows(h, b), sparse_upd(h), (int)h->step_count));
    return SBR_OK;
}
extern "C" int sbr_flush_lazy(sbr_handle* h) {
    CHECK_ARG(h, "null handle");
    return flush_lazy(h);
}

extern "C" int sbr_describe_param(const sbr_config* cfg, int i, char* name, size_t name_cap, int64_t dims[2], int* ndim) {
    CHECK_ARG(cfg && dims && ndim, "null argument");
    Layout lay; std::string err;
    if (sbr_build_layout(*cfg, lay, err) != SBR_OK) { sbr_set_error("%s", err.c_str()); return SBR_EINVAL; }
    std::vector<ParamDesc> descs;
    sbr_param_descs(lay, descs);
    CHECK_ARG(i >= 0 && i < (int)descs.size(), "parameter index %d outside [0,%d)", i, (int)descs.size());
    const ParamDesc& d = descs[i];
    if (name && name_cap) snprintf(name, name_cap, "%s", d.name.c_str());
    dims[0] = d.d0; dims[1] = d.ndim == 2 ? d.d1 : 1; *ndim = d.ndim;
    return SBR_OK;
}

extern "C" int sbr_set_params(sbr_handle* h, int n, const float* const* arrays) {
    CHECK_ARG(h && arrays && n == (int)h->descs.size(), "expected %d parameter arrays, got %d", h ? (int)h->descs.size() : -1, n);
    { const int rc = flush_lazy(h); if (rc != SBR_OK) return rc; }   // pending zero-gradient steps belong to the old values
    std::vector<float> image(h->lay.n_params, 0.0f);     // padding stays exactly zero
    for (int i = 0; i < n; ++i) {
        CHECK_ARG(arrays[i], "parameter array %d is NULL", i);
        convert_param(h->lay, h->descs[i], image.data(), const_cast<float*>(arrays[i]), true);
    }
    SBR_HIP(hipMemcpyAsync(h->P(0), image.data(), image.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    SBR_HIP(hipStreamSynchronize(h->stream));
    h->fwd_done = false;
    return SBR_OK;
}

static int get_section(sbr_handle* h, const float* dev, int n, float* const* arrays) {
    CHECK_ARG(h && arrays && n == (int)h->descs.size(), "expected %d parameter arrays, got %d", h ? (int)h->descs.size() : -1, n);
    std::vector<float> image(h->lay.n_params);
    SBR_HIP(hipMemcpyAsync(image.data(), dev, image.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    SBR_HIP(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; ++i) {
        CHECK_ARG(arrays[i], "parameter array %d is NULL", i);
        convert_param(h->lay, h->descs[i], image.data(), arrays[i], false);
    }
    return SBR_OK;
}
extern "C" int sbr_get_params(sbr_handle* h, int n, float* const* arrays) {
    CHECK_ARG(h, "null handle");
    { const int rc = flush_lazy(h); if (rc != SBR_OK) return rc; }
    return get_section(h, h->P(0), n, arrays);
}
extern "C" int sbr_get_grads(sbr_handle* h, int n, float* const* arrays) { return get_section(h, h ? h->Gd(0) : nullptr, n, arrays); }

extern "C" int sbr_section(sbr_handle* h, int which, void** dev_ptr, size_t* n_floats, size_t* split_floats) {
    CHECK_ARG(h && dev_ptr && n_floats, "null argument");
    const Layout& y = h->lay;
    if (split_floats) *split_floats = y.p_split;
    if (which == 0 || which == 2) { const int rc = flush_lazy(h); if (rc != SBR_OK) return rc; }   // current at the time of the call
    switch (which) {
        case 0: *dev_ptr = h->P(0); *n_floats = y.n_params; return SBR_OK;
        case 1: *dev_ptr = h->Gd(0); *n_floats = y.n_params + 1; return SBR_OK;
        case 2: *dev_ptr = h->St(0, 0); *n_floats = y.n_state_arrays * y.n_params; return SBR_OK;
    }
    sbr_set_error("unknown section %d", which);
    return SBR_EINVAL;
}

// ---------------------------------------------------------------------------------------
// batch
// ---------------------------------------------------------------------------------------
extern "C" int sbr_set_default_target(sbr_handle* h, const float* default_target) {
    CHECK_ARG(h, "null handle");
    const Layout& y = h->lay;
    CHECK_ARG(SBR_LOSS_IS_MARGIN(y.cfg.loss), "only the multi-target losses (hinge / logit / logsig) have a default target");
    if (default_target) SBR_HIP(hipMemcpyAsync(h->A(y.a_dflt), default_target, (size_t)y.N * sizeof(float), hipMemcpyHostToDevice, h->stream));
    else SBR_HIP(hipMemsetAsync(h->A(y.a_dflt), 0, (size_t)y.N * sizeof(float), h->stream));
    SBR_HIP(hipStreamSynchronize(h->stream));
    return SBR_OK;
}

extern "C" int sbr_set_batch(sbr_handle* h, const int32_t* X, const int32_t* lengths, const int32_t* target,
                             const int32_t* samples, const float* pop, int n_rows, int on_device) {
    CHECK_ARG(h && X && lengths, "null X / lengths");
    const Layout& y = h->lay;
    CHECK_ARG(n_rows >= 1 && n_rows <= y.B, "n_rows %d outside [1,%d]", n_rows, y.B);
    const bool margin = SBR_LOSS_IS_MARGIN(y.cfg.loss);
    const int n_tgt = y.S > 0 ? y.Bg : n_rows * y.NT;
    if (!on_device) {   // the reference would raise IndexError inside Theano for bad ids; check on host
        for (size_t i = 0; i < (size_t)n_rows * y.T * y.F; ++i)
            CHECK_ARG(X[i] >= 0 && X[i] < y.cfg.input_size, "input index %d out of range [0,%d)", X[i], y.cfg.input_size);
        for (int i = 0; i < n_rows; ++i) CHECK_ARG(lengths[i] >= 0 && lengths[i] <= y.T, "length %d outside [0,%d]", lengths[i], y.T);
        if (target) for (int i = 0; i < n_tgt; ++i) CHECK_ARG((target[i] >= 0 || (margin && target[i] == -1)) && target[i] < y.N, "target %d out of range", target[i]);
        if (samples) for (int i = 0; i < y.S; ++i) CHECK_ARG(samples[i] >= 0 && samples[i] < y.N, "sample %d out of range", samples[i]);
    }
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    hipStream_t s = h->stream;
    h->bX = (const int*)h->A(y.a_X); h->blen = (const int*)h->A(y.a_len); h->btgt = (const int*)h->A(y.a_tgt);
    h->bsmp = (const int*)h->A(y.a_smp); h->bpop = h->A(y.a_pop);
    h->bb_set = 0; h->bb_unread = false;      // (written on the main stream, behind every reader of the set: sbr_build_batch's stream was joined when it built)
    if (on_device && n_rows == y.Bp && (pop || margin)) {
        // device-resident inputs that cover every (padded) row: use them in place, no copies.  The caller
        // keeps them alive and unchanged until the step has run (stream order), as with any device input.
        h->bX = X; h->blen = lengths; if (pop) h->bpop = pop;
        if (target) h->btgt = target;
        if (samples && y.S > 0) h->bsmp = samples;
    } else {
        if (n_rows < y.Bp) {   // padded rows: index 0, length 0, popularity 1
            if (margin) SBR_HIP(hipMemsetAsync(h->A(y.a_tgt), 0xFF, (size_t)y.Bp * y.NT * sizeof(int), s));   // no positives
            SBR_HIP(hipMemsetAsync(h->A(y.a_X), 0, (size_t)y.Bp * y.T * y.F * sizeof(int), s));
            SBR_HIP(hipMemsetAsync(h->A(y.a_len), 0, (size_t)y.Bp * sizeof(int), s));
            SBR_LAUNCH(launch_fill(s, h->A(y.a_pop), 1.0f, y.Bp));
        }
        SBR_HIP(hipMemcpyAsync(h->A(y.a_X), X, (size_t)n_rows * y.T * y.F * sizeof(int), kind, s));
        SBR_HIP(hipMemcpyAsync(h->A(y.a_len), lengths, (size_t)n_rows * sizeof(int), kind, s));
        if (target) SBR_HIP(hipMemcpyAsync(h->A(y.a_tgt), target, (size_t)n_tgt * sizeof(int), kind, s));
        if (samples && y.S > 0) SBR_HIP(hipMemcpyAsync(h->A(y.a_smp), samples, (size_t)y.S * sizeof(int), kind, s));
        if (pop) SBR_HIP(hipMemcpyAsync(h->A(y.a_pop), pop, (size_t)n_rows * sizeof(float), kind, s));
        else SBR_LAUNCH(launch_fill(s, h->A(y.a_pop), 1.0f, y.Bp));
    }
    if (!on_device) SBR_HIP(hipStreamSynchronize(s));   // caller's host arrays may be freed on return
    h->n_rows = n_rows; h->have_batch = true; h->fwd_done = false;
    return SBR_OK;
}

// ---------------------------------------------------------------------------------------
// step phases
// ---------------------------------------------------------------------------------------
static RecArgs rec_args(sbr_handle* h, int l) {
    const Layout& y = h->lay; const LayerLayout& ly = y.layer[l];
    RecArgs a; memset(&a, 0, sizeof(a));
    a.cell = y.cfg.cell; a.T = y.T; a.Bp = y.Bp; a.H = ly.H; a.Hp = ly.Hp; a.G = y.G;
    a.n_in = ly.n_in_p;
    a.clip = y.cfg.grad_clip;
    a.len = h->blen;
    a.xt = h->A(ly.a_xt); a.Whid = h->P(ly.p_Whid); a.peep = h->P(ly.p_peep);
    a.cinit = h->P(ly.p_cinit); a.hinit = h->P(ly.p_hinit);
    a.hs = h->A(ly.a_hs); a.cs = h->A(ly.a_cs);
    if (ly.Hp == 256 || ly.Hp == 512) { a.xh = h->A(ly.a_xh); a.pring = h->A(ly.a_pring); }
    for (int k = 0; k < 4; ++k) a.g[k] = h->A(ly.a_g[k]);
    a.dxt = h->A(ly.a_dxt); a.dhi = h->A(ly.a_dhi); a.part = h->A(ly.a_part);
    a.xt_blocked = 0;
    a.rpt = h->rpt; a.x6_split = h->x6_split; a.x6_pipe = h->x6_pipe;
    a.t_lo = 0; a.t_hi = y.T; a.chunk = 0; a.state = h->A(ly.a_state);
    a.f32_mfma = (y.cfg.flags & SBR_FLAG_F32_MFMA) ? 1 : 0;
    a.prof = (y.cfg.flags & SBR_FLAG_PROFILE_REC) ? (unsigned long long*)h->A(y.a_prof) : nullptr;
    a.cluster = h->cluster; a.cl_linear = h->cl_linear; a.fault = (int*)h->A(y.a_fault);
    a.clx = (int*)h->A(y.a_clx); a.epoch = (++h->cl_epoch) & 0x07FFFFFF;
    a.relu = (y.cfg.cell == SBR_CELL_VANILLA && (l / y.D > 0 || y.E)) ? 1 : 0;   // stock RecurrentLayer: rectify [3P]
    return a;
}
static inline bool simple_rec(const sbr_handle* h) { return h->lay.cfg.flags & SBR_FLAG_SIMPLE_REC; }
// Operand planes of the dense GEMMs around the recurrent layers (layer >= 2 input projection and its backward pair, the logits):
// their operands are hidden states in [-1, 1] (not behind a rectifier), weights, and -- in the backward pair -- gate gradients
// that have passed the reference's clip at +-100 (recurrent_layers.py:19): the two-plane fp16 split, three MFMAs per product, f32-class
// (SBR_GEMM_F16=0: bf16x6 as in rounds 1-3).  SBR_FLAG_BF16_LAYERS: plain bf16 operands, one MFMA (BASELINE configs[4]).
static inline bool layer_gemm_f16(const sbr_handle* h, bool with_gradient) {
    const Layout& y = h->lay;
    static const int on = [] { const char* e = getenv("SBR_GEMM_F16"); return e ? atoi(e) : 1; }();
    if (!on || y.cfg.cell == SBR_CELL_VANILLA) return false;
    return !with_gradient || (y.cfg.grad_clip > 0.0f && y.cfg.grad_clip <= 100.0f);
}
static inline void layer_gemm_hint(const sbr_handle* h, bool grad_a, bool grad_b) {
    if (h->lay.cfg.flags & SBR_FLAG_BF16_LAYERS) sbr_gemm_hint(1, 1.0f, 1.0f);
    else if (layer_gemm_f16(h, grad_a || grad_b)) sbr_gemm_hint(2, grad_a ? 512.0f : 1.0f, grad_b ? 512.0f : 1.0f);
}
static inline bool simple_gemm(const sbr_handle* h) { return h->lay.cfg.flags & SBR_FLAG_SIMPLE_GEMM; }
// Overlapped step tail: time chunks for this step (0 = not taken) and steps per chunk.  Taken for a single index-input layer
// served by rec_bwd_x6p's progress-publishing form, dense updates, the bf16x6 weight-gradient GEMM and one BPTT launch.
static int tail_plan(sbr_handle* h, int* ch_out) {
    const Layout& y = h->lay;
    *ch_out = 0;
    if (!h->tail_overlap || y.tail_keys < 2 || y.n_sparse || h->bwd_chunks != 1 || !h->wgrad_x6) return 0;
    if (simple_rec(h) || simple_gemm(h) || (y.cfg.flags & (SBR_FLAG_F32_MFMA | SBR_FLAG_ATOMIC_SCATTER))) return 0;
    RecArgs a; memset(&a, 0, sizeof(a));
    a.cell = y.cfg.cell; a.T = y.T; a.Bp = y.Bp; a.H = y.layer[0].H; a.Hp = y.layer[0].Hp; a.G = y.G; a.clip = y.cfg.grad_clip;
    a.rpt = h->rpt; a.x6_split = h->x6_split; a.x6_pipe = h->x6_pipe; a.n_in = y.layer[0].n_in_p;
    a.hs = h->A(y.layer[0].a_hs); a.cs = h->A(y.layer[0].a_cs);
    for (int k = 0; k < 4; ++k) a.g[k] = h->A(y.layer[0].a_g[k]);
    if (!sbr_rec_x6p_tail_ok(a)) return 0;
    int nc = std::min(std::min(y.tail_keys, h->tail_chunks_max), y.T / 16);
    if (nc < 2) return 0;
    const int ch = (y.T + nc - 1) / nc;
    nc = (y.T + ch - 1) / ch;
    if (nc < 2) return 0;
    *ch_out = ch;
    // Chunk bounds.  The scatter-add of a time chunk can start when the chain has left it, and the chain leaves chunk 0 last:
    // with equal chunks an eighth of the step's entries waits for the chain's end (30 - 45 us of polling waves behind it,
    // profiles/round3_c_timeline.txt).  So the chunks near t = 0 are small -- 1, 3, 7, 18 ... steps (powers of SBR_TAIL_GEOM,
    // default 2.6; <= 1: equal chunks) -- until a power exceeds the equal share of what is left, which the remaining chunks then
    // take: at T = 200 and eight chunks 1, 3, 7, 18, 42, 43, 43, 43 steps (the consumers still start after a fifth of the chain).
    // At most half of the chunks are small ones.
    const double geom = h->tail_geom;
    SbrTChunks& tc = h->tail_bounds;
    tc.n = nc;
    tc.lo[0] = 0;
    // LDS-row scatter-add (launch_scatter_lds_poll): a unit walks the chunks one after the other, so what counts is that chunk c is
    // done when chunk c - 1 is released and that ONE round of rows is left behind the chain: the last chunk has SBR_TAIL_FIRST
    // steps (default 6: ~8 entries per unit) and the sizes grow by SBR_TAIL_GEOM (default 1.6 here): 6, 10, 15, 25, then equal shares.
    double pw = h->tail_scatter_lds ? (double)h->tail_first : 1.0;
    for (int c = 0; c < nc; ++c) {
        const int rem = y.T - tc.lo[c], left = nc - c;
        const int share = (rem + left - 1) / left;
        const int n_small = nc >= 4 ? nc / 2 : (nc - 1) / 2;
        int sz = (geom > 1.0 && c < n_small) ? std::min(share, std::max(1, (int)(pw + 0.5))) : (geom > 1.0 ? share : std::min(rem, ch));
        if (c == nc - 1) sz = rem;
        sz = std::max(1, std::min(sz, rem - (left - 1)));            // every later chunk keeps at least one step
        tc.lo[c + 1] = tc.lo[c] + sz;
        pw *= geom > 1.0 ? geom : 1.0;
    }
    for (int c = nc; c <= SBR_TCHUNKS_MAX; ++c) tc.lo[c] = y.T;
    return nc;
}

// behind the time-chunked sort of an overlapped tail (second side stream): the running cost of the ids, for the LDS-row scatter-add
static int tail_cost_scan(sbr_handle* h) {
    const Layout& y = h->lay;
    h->tail_cost_scanned = false;
    if (!h->tail_scatter_lds) return SBR_OK;
    hipError_t e = hipSuccess;
    if (launch_scatter_cost_scan(h->side2, (const int*)h->A(y.a_soff), (int*)h->A(y.a_sP), y.cfg.input_size, h->tail_nc, y.T * y.Bp * y.F,
                                 y.G * y.layer[0].Hp, h->tail_scatter_units, &e)) {
        if (e != hipSuccess) { sbr_set_error("HIP launch failed: %s", hipGetErrorString(e)); return SBR_EHIP; }
        h->tail_cost_scanned = true;
    }
    return SBR_OK;
}

// sbr_chain_times: events around one chain launch (dir 0 = forward, 1 = backward); `which` 0 in front of it, 1 behind it
static inline void chain_mark(sbr_handle* h, hipStream_t st, int dir, int which) {
    if (!h->chain_timing || h->ch_n >= sbr_handle::kChain) return;
    hipEvent_t& e = h->ev_ch[h->ch_n][which];
    if (!e && hipEventCreate(&e) != hipSuccess) { e = nullptr; return; }
    (void)hipEventRecord(e, st);
    if (which == 1) { h->ch_dir[h->ch_n] = (unsigned char)dir; h->ch_n += 1; }
}
#define SBR_LAUNCH_CHAIN(DIR, STREAM, CALL) do { chain_mark(h, STREAM, DIR, 0); SBR_LAUNCH(CALL); chain_mark(h, STREAM, DIR, 1); } while (0)

static inline void mark_on(sbr_handle* h, int i, hipStream_t st) {
    if (i == 0) h->marks_shared = 0;
    if ((h->marks_shared >> i) & 1) return;              // this step's mark i was recorded by record_shared
    if (h->timing && ((h->timing_marks >> i) & 1) && h->ev[h->ring_cur][i]) (void)hipEventRecord(h->ev[h->ring_cur][i], st);
}
static inline void mark(sbr_handle* h, int i) { mark_on(h, i, h->stream); }
// An event record costs the stream ~6 us before its next kernel starts (measured: profiles/round1_i_timeline.txt), so
// where a cross-stream event and a timing mark fall on the same point of the main stream ONE record serves both: the
// side stream waits on the timing event.  Returns the event to wait on.
static inline hipEvent_t record_shared(sbr_handle* h, hipEvent_t plain, int mk) {
    if (mk >= 0 && h->timing && ((h->timing_marks >> mk) & 1) && h->ev[h->ring_cur][mk]) {
        (void)hipEventRecord(h->ev[h->ring_cur][mk], h->stream);
        h->marks_shared |= 1u << mk;
        return h->ev[h->ring_cur][mk];
    }
    (void)hipEventRecord(plain, h->stream);
    return plain;
}

extern "C" int sbr_zero_grads(sbr_handle* h) {
    CHECK_ARG(h, "null handle");
    h->step_open = true;                          // a training step begins: sbr_forward may start its batch-only work
    h->out_early = false; h->dh_slabs_n = 0;
    if (!h->in_train_step && h->timing) {         // phase-by-phase step (data-parallel driver): this call opens the step
        h->ring_cur = h->ring_used % sbr_handle::kRing;
        mark(h, 0);
    }
    if (h->grads_clean) return SBR_OK;            // the optimizer kernel zeroes every gradient it consumes
    SBR_HIP(hipMemsetAsync(h->Gd(0), 0, (h->lay.n_params + 1) * sizeof(float), h->stream));
    h->grads_clean = true;
    return SBR_OK;
}

// ---------------------------------------------------------------------------------------
// --r_bi (recurrent_layers.py:70-76).  Level l = forward layer lay.layer[2l] + backwards layer lay.layer[2l+1] over the
// same input.  The backwards layer runs the ordinary kernels on per-row time-reversed copies of its input (see the
// helper kernels in sbr_misc.hip), so every recurrent kernel of the unidirectional path is reused unchanged.
// ---------------------------------------------------------------------------------------
static int side_join(sbr_handle* h);

static int forward_bi(sbr_handle* h) {
    const Layout& y = h->lay; hipStream_t s = h->stream;
    const int TB = y.T * y.Bp;
    int* Xr = (int*)h->A(y.a_Xr);
    if (!y.E) SBR_LAUNCH(launch_rev_rows_int(s, h->bX, h->blen, Xr, y.T, y.Bp, y.F));
    else {
        SBR_LAUNCH(launch_gather_concat(s, h->P(y.p_Emb), h->bX, h->A(y.a_emb), y.T, y.Bp, y.F, y.Ep));
        SBR_LAUNCH(launch_rev_rows(s, h->A(y.a_emb), h->blen, h->A(y.a_embr), y.T, y.Bp, y.F * y.Ep));
    }
    for (int l = 0; l < y.L; ++l) {
        for (int d = 0; d < 2; ++d) {
            const int pl = 2 * l + d;
            const LayerLayout& ly = y.layer[pl];
            const int GHp = y.G * ly.Hp;
            RecArgs ra = rec_args(h, pl);
            if (l == 0 && !y.E) {
                const int* idx = d ? Xr : h->bX;
                if (y.F == 1 && h->fuse_gather && sbr_rec_fwd_can_fuse_gather(ra, simple_rec(h))) {
                    ra.gX = idx; ra.gWin = h->P(ly.p_Win); ra.gbias = h->P(ly.p_b);
                } else {
                    SBR_LAUNCH(launch_gather_xt(s, h->P(ly.p_Win), h->P(ly.p_b), idx, h->A(ly.a_xt), y.T, y.Bp, y.F, GHp, h->n_rows));
                }
            } else {   // dense input: the (reversed) flattened embeddings or the (reversed) concatenated outputs of the level below
                const float* inp = l == 0 ? h->A(d ? y.a_embr : y.a_emb) : h->A(d ? y.a_catr[l - 1] : y.a_cat[l - 1]);
                SBR_LAUNCH(launch_gemm(s, inp, ly.n_in_p, 1, h->P(ly.p_Win), GHp, 1, h->A(ly.a_xt), GHp, TB, GHp, ly.n_in_p,
                                       h->P(ly.p_b), nullptr, 0, simple_gemm(h)));
            }
            if (l == 0 && d == 1) mark(h, 1);
            SBR_LAUNCH_CHAIN(0, s, launch_rec_forward(s, ra, simple_rec(h)));
        }
        const LayerLayout& lf = y.layer[2 * l]; const LayerLayout& lb = y.layer[2 * l + 1];
        if (l + 1 < y.L) {
            SBR_LAUNCH(launch_cat_outputs(s, h->A(lf.a_hs), h->A(lb.a_hs), h->blen, h->A(y.a_cat[l]), y.T, y.Bp, lf.Hp));
            SBR_LAUNCH(launch_rev_rows(s, h->A(y.a_cat[l]), h->blen, h->A(y.a_catr[l]), y.T, y.Bp, 2 * lf.Hp));
        } else {   // only_return_final: both directions' last scan output (sparse_lstm.py:485-486)
            SBR_LAUNCH(launch_hcat(s, h->A(lf.a_hs) + (size_t)TB * lf.Hp, h->A(lb.a_hs) + (size_t)TB * lb.Hp, h->A(y.a_hcat), y.Bp, lf.Hp));
        }
    }
    mark(h, 2);
    h->fwd_done = true;
    return SBR_OK;
}

static int backward_bi(sbr_handle* h) {
    const Layout& y = h->lay; hipStream_t s = h->stream;
    const bool sg = simple_gemm(h);
    float* ws = h->A(y.a_ws);
    const int TB = y.T * y.Bp;
    SBR_LAUNCH(launch_split_cols(s, h->A(y.a_dhlast), h->A(y.a_dhl[0]), h->A(y.a_dhl[1]), y.Bp, y.HLp));
    for (int l = y.L - 1; l >= 0; --l) {
        for (int d = 0; d < 2; ++d) {
            const int pl = 2 * l + d;
            const LayerLayout& ly = y.layer[pl];
            const int GHp = y.G * ly.Hp;
            RecArgs a = rec_args(h, pl);
            if (h->fill_done && sbr_rec_cluster_ok(a)) {
                a.sentinel_done = 1;
                SBR_HIP(hipStreamWaitEvent(s, h->ev_fill, 0));
            }
            a.dh_last = l == y.L - 1 ? h->A(y.a_dhl[d]) : nullptr;
            a.dh_ext = l < y.L - 1 ? h->A(ly.a_dhext) : nullptr;
            const int nblk = sbr_rec_bwd_blocks(a, simple_rec(h));
            SBR_LAUNCH_CHAIN(1, s, launch_rec_backward(s, a, simple_rec(h)));
            if (l == 0 && d == 1) mark(h, 4);
            SBR_LAUNCH(launch_rec_reduce_partials(s, a.part, nblk, y.G, ly.Hp, y.cfg.cell, h->Gd(ly.p_b), h->Gd(ly.p_peep),
                                                  h->Gd(ly.p_cinit), h->Gd(ly.p_hinit)));
            // dW_hid = hs^T . d hid_input (hs slot t = the state before step t)
            if (y.cfg.cell == SBR_CELL_GRU) {
                SBR_LAUNCH(launch_gemm(s, h->A(ly.a_hs), 1, ly.Hp, a.dxt, GHp, 1, h->Gd(ly.p_Whid), GHp, ly.Hp, 2 * ly.Hp, TB, nullptr,
                                       ws, y.ws_floats, sg));
                SBR_LAUNCH(launch_gemm(s, h->A(ly.a_hs), 1, ly.Hp, a.dhi, ly.Hp, 1, h->Gd(ly.p_Whid) + 2 * ly.Hp, GHp, ly.Hp, ly.Hp, TB,
                                       nullptr, ws, y.ws_floats, sg));
            } else {
                SBR_LAUNCH(launch_gemm(s, h->A(ly.a_hs), 1, ly.Hp, a.dxt, GHp, 1, h->Gd(ly.p_Whid), GHp, ly.Hp, GHp, TB, nullptr, ws,
                                       y.ws_floats, sg));
            }
            if (l == 0 && !y.E) {   // index input: scatter-add with this direction's ids (the backwards one sorted its reversed ids)
                if (d == 0) mark(h, 5);
                const int* idx = d ? (const int*)h->A(y.a_Xr) : h->bX;
                if (y.cfg.flags & SBR_FLAG_ATOMIC_SCATTER) {
                    SBR_LAUNCH(launch_scatter_rows(s, h->Gd(ly.p_Win), a.dxt, idx, a.len, y.T, y.Bp, y.F, GHp));
                } else {
                    SBR_HIP(hipStreamWaitEvent(s, h->ev_sort, 0));
                    SBR_LAUNCH(launch_scatter_reduce(s, h->Gd(ly.p_Win), a.dxt, (const int*)h->A(d ? y.a_s2sid : y.a_sid),
                                                     (const int*)h->A(d ? y.a_s2pos : y.a_spos), (const int*)h->A(d ? y.a_s2off : y.a_soff),
                                                     y.cfg.input_size, y.T * y.Bp * y.F, GHp, y.Bp));
                }
            } else {                // dense input: dW_in = inp^T . dxt, d_inp = dxt . W_in^T (in this direction's time order)
                const float* inp = l == 0 ? h->A(d ? y.a_embr : y.a_emb) : h->A(d ? y.a_catr[l - 1] : y.a_cat[l - 1]);
                SBR_LAUNCH(launch_gemm(s, inp, 1, ly.n_in_p, a.dxt, GHp, 1, h->Gd(ly.p_Win), GHp, ly.n_in_p, GHp, TB, nullptr, ws,
                                       y.ws_floats, sg));
                SBR_LAUNCH(launch_gemm(s, a.dxt, GHp, 1, h->P(ly.p_Win), 1, GHp, h->A(y.a_dinp[d]), ly.n_in_p, TB, ly.n_in_p, GHp, nullptr,
                                       nullptr, 0, sg));
            }
        }
        if (l > 0) {        // gradient wrt the level below: forward half in forward time, backwards half in reversed time
            const LayerLayout& lf = y.layer[2 * (l - 1)]; const LayerLayout& lb = y.layer[2 * (l - 1) + 1];
            SBR_LAUNCH(launch_uncat(s, h->A(y.a_dinp[0]), h->A(y.a_dinp[1]), h->blen, h->A(lf.a_dhext), h->A(lb.a_dhext), y.T, y.Bp,
                                    2 * lf.Hp, lf.Hp));
        } else if (y.E) {   // embedding table: both directions' input gradients, back in forward time, scatter-added by index
            mark(h, 5);
            SBR_LAUNCH(launch_uncat(s, h->A(y.a_dinp[0]), h->A(y.a_dinp[1]), h->blen, h->A(y.a_demb), nullptr, y.T, y.Bp, y.F * y.Ep, 0));
            SBR_HIP(hipStreamWaitEvent(s, h->ev_sort, 0));
            SBR_LAUNCH(launch_scatter_reduce(s, h->Gd(y.p_Emb), h->A(y.a_demb), (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos),
                                             (const int*)h->A(y.a_soff), y.cfg.input_size, y.T * y.Bp * y.F, y.Ep, y.Bp));
        }
        if (l == 0) mark(h, 6);
    }
    if (!h->in_train_step && !h->deferred_join) return side_join(h);
    return SBR_OK;
}

extern "C" int sbr_forward(sbr_handle* h) {
    CHECK_ARG(h, "null handle");
    sbr_gemm_set_exact_f32((h->lay.cfg.flags & SBR_FLAG_F32_MFMA) != 0);
    if (!h->have_batch) { sbr_set_error("sbr_forward: no batch set"); return SBR_ESTATE; }
    const Layout& y = h->lay; hipStream_t s = h->stream;
    if (sparse_lazy(h))      // the rows this batch gathers must be current before they are read
        for (int b = 0; b < y.n_sparse; ++b)
            if (y.sparse[b].kind == 0)
                SBR_LAUNCH(launch_sparse_catch_up_batch(s, sparse_rows(h, b), sparse_upd(h), h->bX, h->blen, y.T, y.Bp, y.F, (int)h->step_count));
    h->tail_nc = h->step_open ? tail_plan(h, &h->tail_ch) : 0;      // overlapped tail for this step? (never for predict / top-k)
    const bool training = h->step_open;
    h->step_open = false;
    h->tail_sorted = false;
    // (sbr_build_batch fills the batch set this forward does not read while the step runs: what it needs to know to do that safely)
    h->set_use[h->bb_set] = ++h->batch_seq; h->bb_unread = false;
    if (training) { if (h->train_fwd_open) h->bb_slow = 2; h->train_fwd_open = true; }
    // Sampled heads with lazily stepped W_out rows: which cells the step samples depends on the batch only, and catching their
    // rows up is a chain of dependent replays per row (C3: 81 us, C5: 94 us for 288 rows) that sat on the main stream between the
    // forward chain and the head.  It runs now on the side stream, beside the forward chain; sbr_loss_backward_output waits for
    // its event (long complete by then).  SBR_SPARSE_OUT_EARLY=0: as before.
    h->cells_early = false;
    bool forked = false;
    if (training && h->sparse_out_early && sparse_lazy(h) && y.S > 0 && y.cfg.loss != SBR_LOSS_CCE && !SBR_LOSS_IS_MARGIN(y.cfg.loss)) {
        int kb = -1;
        for (int b = 0; b < y.n_sparse; ++b) if (y.sparse[b].kind == 1) kb = b;
        if (kb >= 0) {
            SBR_HIP(hipEventRecord(h->ev_fork, s)); forked = true;      // (device-resident batches are produced on the main stream)
            SBR_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
            int* cells = (int*)h->A(y.a_cells);
            SBR_LAUNCH(launch_build_cells(h->side, h->btgt, h->bsmp, y.Bg, y.S, cells));
            SBR_LAUNCH(launch_sparse_catch_up_list(h->side, sparse_rows(h, kb), sparse_upd(h), cells, nullptr, y.C, y.C, (int)h->step_count));
            SBR_HIP(hipEventRecord(h->ev_cells, h->side));
            h->cells_early = true;
            h->side_pending = true;      // (parameters, optimizer state and last[] were written over there: a step abandoned behind
                                         // sbr_forward -- an error return, a ranking, an export -- joins before it reads them)
        }
    }
    // Overlapped tail, round 3.  Its consumers are throughput-bound once they have the chip's other 192 CUs to themselves (the
    // fence below), so WHEN they start decides when the step ends -- and both waited behind work that does not need the chain:
    // the scatter-add behind the 45 us of the time-chunked sort.  The sort needs nothing but the batch: it runs now, beside the
    // forward chain, for one event record in front of it.  The forward chain claims its CUs' LDS while the sort (118 KB of LDS
    // histogram per workgroup) runs beside it, so the two do not share CUs.  SBR_TAIL_EARLY_SORT=0: behind the output phase.
    const bool tail_live = h->tail_nc >= 2 && h->tail_overlap == 1;
    if (tail_live && h->tail_early_sort) {
        if (!forked) SBR_HIP(hipEventRecord(h->ev_fork, s));
        SBR_HIP(hipStreamWaitEvent(h->side2, h->ev_fork, 0));
        SBR_LAUNCH(launch_scatter_sort(h->side2, h->bX, h->blen, y.T, y.Bp, y.F, y.cfg.input_size, (int*)h->A(y.a_scnt),
                                       (int*)h->A(y.a_soff), (int*)h->A(y.a_scur), (int*)h->A(y.a_sid), (int*)h->A(y.a_spos), 0,
                                       h->tail_ch, h->tail_nc, &h->tail_bounds, &h->scnt_zero_n));
        h->tail_sorted = true;
        { const int rc = tail_cost_scan(h); if (rc != SBR_OK) return rc; }
    }
    if (y.D == 2) return forward_bi(h);
    for (int l = 0; l < y.L; ++l) {
        const LayerLayout& ly = y.layer[l];
        const int GHp = y.G * ly.Hp;
        RecArgs ra = rec_args(h, l);
        if (l == 0 && y.E) {   // --r_emb: embeddings of the F indices, flattened, then a dense input projection
            SBR_LAUNCH(launch_gather_concat(s, h->P(y.p_Emb), h->bX, h->A(y.a_emb), y.T, y.Bp, y.F, y.Ep));
            SBR_LAUNCH(launch_gemm(s, h->A(y.a_emb), ly.n_in_p, 1, h->P(ly.p_Win), GHp, 1, h->A(ly.a_xt), GHp, y.T * y.Bp, GHp,
                                   ly.n_in_p, h->P(ly.p_b), nullptr, 0, simple_gemm(h)));
            mark(h, 1);
        } else if (l == 0) {
            if (y.F == 1 && h->fuse_gather && sbr_rec_fwd_can_fuse_gather(ra, simple_rec(h))) {
                ra.gX = h->bX; ra.gWin = h->P(ly.p_Win); ra.gbias = h->P(ly.p_b);   // gathered inside the forward kernel
            } else {
                SBR_LAUNCH(launch_gather_xt(s, h->P(ly.p_Win), h->P(ly.p_b), h->bX, h->A(ly.a_xt), y.T, y.Bp,
                                            y.F, GHp, h->n_rows));
            }
            mark(h, 1);
        } else {   // dense layers: xt = hid_out(l-1) . W_in + b  (Lasagne precompute_input [3P], recurrent_layers.py:94-104)
            const LayerLayout& lo = y.layer[l - 1];
            layer_gemm_hint(h, false, false);
            SBR_LAUNCH(launch_gemm(s, h->A(lo.a_hs) + (size_t)y.Bp * lo.Hp, lo.Hp, 1, h->P(ly.p_Win), GHp, 1, h->A(ly.a_xt), GHp,
                                   y.T * y.Bp, GHp, lo.Hp, h->P(ly.p_b), nullptr, 0, simple_gemm(h)));
        }
        if (l == 0 && h->tail_sorted) ra.fence_kb = h->tail_fence_kb;
        SBR_LAUNCH_CHAIN(0, s, launch_rec_forward(s, ra, simple_rec(h)));
    }
    mark(h, 2);
    h->fwd_done = true;
    return SBR_OK;
}

static float* h_last(sbr_handle* h) {   // hid_out[-1] (sparse_lstm.py:485-486) = slot T of the top layer
    const Layout& y = h->lay; const LayerLayout& ly = y.layer[(y.L - 1) * y.D];
    if (y.D == 2) return h->A(y.a_hcat);                 // [forward final | backwards final], filled by forward_bi
    return h->A(ly.a_hs) + (size_t)y.T * y.Bp * ly.Hp;
}

// The side stream carries everything that only feeds the optimizer (output-layer weight/bias gradients, the cost
// scalar, the counting sort for the embedding scatter, the weight-gradient GEMM of finished BPTT chunks) so that
// the main stream holds nothing but the dependent chain  logits -> softmax -> dh -> BPTT chunks -> scatter.
static int side_join(sbr_handle* h) {
    if (h->tail_join_pending) {      // overlapped tail of a phase-by-phase step: both consumer streams
        SBR_HIP(hipStreamWaitEvent(h->stream, h->ev_tail2, 0));
        SBR_HIP(hipStreamWaitEvent(h->stream, h->ev_tail, 0));
        h->tail_join_pending = false; h->side_pending = false;
    }
    if (h->side_pending) {
        SBR_HIP(hipEventRecord(h->ev_join, h->side));
        SBR_HIP(hipStreamWaitEvent(h->stream, h->ev_join, 0));
        h->side_pending = false;
    }
    return SBR_OK;
}

extern "C" int sbr_loss_backward_output(sbr_handle* h) {
    CHECK_ARG(h, "null handle");
    sbr_gemm_set_exact_f32((h->lay.cfg.flags & SBR_FLAG_F32_MFMA) != 0);
    if (!h->fwd_done) { sbr_set_error("sbr_loss_backward_output: call sbr_forward first"); return SBR_ESTATE; }
    const Layout& y = h->lay; hipStream_t s = h->stream, sd = h->side;
    const int R = h->n_rows, Hp = y.HLt, N = y.N;      // Hp: the output layer's input width (both directions with --r_bi)
    const bool sg = simple_gemm(h);
    h->grads_clean = false;
    h->dh_slabs_n = 0;
    float* hl = h_last(h);
    float* ws = h->A(y.a_ws);
    float* ws2 = h->A(y.a_ws2);
    const int* tgt = h->btgt;
    if (R < y.Bp) SBR_HIP(hipMemsetAsync(h->A(y.a_dhlast), 0, (size_t)y.Bp * Hp * sizeof(float), s));   // padded rows carry no gradient
    h->side_pending = true;
    h->fill_done = false;
    h->out3 = false;
    // Work on the side stream that needs only the batch: the sentinel fill of the cluster BPTT kernels' exchange arrays and
    // the sort for the embedding scatter-add (the scatter kernel waits for ev_sort).  With cluster kernels it starts now,
    // beside the output phase (its own fork event); otherwise it rides behind the ev_lg wait the side stream needs anyway
    // -- every event record costs the main stream a few microseconds.
    bool fill_needed = false;
    if (!simple_rec(h))
        for (int l = 0; l < y.L * y.D; ++l) fill_needed = fill_needed || sbr_rec_cluster_ok(rec_args(h, l));
    auto side_batch_work = [&]() -> int {
        if (fill_needed) {
            for (int l = 0; l < y.L * y.D; ++l) {
                RecArgs a = rec_args(h, l);
                if (sbr_rec_cluster_ok(a)) SBR_LAUNCH(sbr_rec_bwd_cl_fill(sd, a));
            }
            SBR_HIP(hipEventRecord(h->ev_fill, sd)); h->fill_done = true;
        }
        // (round 6, call s3: the sort BEHIND the head's record instead of beside the head lets the one-launch sampled head run in 26 us
        // instead of 24 .. 72 by workgroup -- the sort's counting kernels are all atomics -- but beside the BPTT chain it costs the chain
        // more: rec_bwd_c16 375 -> 441 us at C3, 390 -> 404 at C4.  It stays here.)
        if (h->tail_nc >= 2) {
            // overlapped tail: the time-chunked sort runs on the SECOND side stream, which consumes it (scatter-add beside the
            // chain); that stream is released by the same record as the first one
            if (h->ev_lg_rec) SBR_HIP(hipStreamWaitEvent(h->side2, h->ev_lg_rec, 0));      // (NULL: the head's flag form -- the sort ran
                                                                                        // beside the forward chain, the consumers wait for the chain's progress words)
            if (!h->tail_sorted) {
                SBR_LAUNCH(launch_scatter_sort(h->side2, h->bX, h->blen, y.T, y.Bp, y.F, y.cfg.input_size, (int*)h->A(y.a_scnt),
                                               (int*)h->A(y.a_soff), (int*)h->A(y.a_scur), (int*)h->A(y.a_sid), (int*)h->A(y.a_spos), 0,
                                               h->tail_ch, h->tail_nc, &h->tail_bounds, &h->scnt_zero_n));
                { const int rc = tail_cost_scan(h); if (rc != SBR_OK) return rc; }
            }
        } else if (!(y.cfg.flags & SBR_FLAG_ATOMIC_SCATTER) || y.E || y.n_sparse) {
            SBR_LAUNCH(launch_scatter_sort(sd, h->bX, h->blen, y.T, y.Bp, y.F,
                                           y.cfg.input_size, (int*)h->A(y.a_scnt), (int*)h->A(y.a_soff), (int*)h->A(y.a_scur),
                                           (int*)h->A(y.a_sid), (int*)h->A(y.a_spos), y.E ? 1 : 0, 0, 1, nullptr, &h->scnt_zero_n));
            if (y.D == 2 && !y.E)   // the backwards direction scatters with the reversed ids (a_Xr was written by forward_bi)
                SBR_LAUNCH(launch_scatter_sort(sd, (const int*)h->A(y.a_Xr), h->blen, y.T, y.Bp, y.F, y.cfg.input_size,
                                               (int*)h->A(y.a_s2cnt), (int*)h->A(y.a_s2off), (int*)h->A(y.a_s2cur),
                                               (int*)h->A(y.a_s2sid), (int*)h->A(y.a_s2pos), 0));
            SBR_HIP(hipEventRecord(h->ev_sort, sd));
        }
        return SBR_OK;
    };
    if (fill_needed) {
        SBR_HIP(hipEventRecord(h->ev_fork, s));
        SBR_HIP(hipStreamWaitEvent(sd, h->ev_fork, 0));
        const int rc = side_batch_work(); if (rc != SBR_OK) return rc;
    }
    if (y.cfg.loss == SBR_LOSS_CCE || SBR_LOSS_IS_MARGIN(y.cfg.loss)) {      // dense heads: full softmax, or RNNMargin's linear layer
        float* lg = h->A(y.a_logits);
        const int Nl = (N + 3) & ~3;               // row stride of the logits / dlogits buffer
        // logits = h . W_out (+ b inside the softmax kernel): DenseLayer (rnn_one_hot.py:65)
        const bool bf16p = (y.cfg.flags & SBR_FLAG_BF16_PROJECTION) && !sg;
        // critical path: dh = dlogits . W_out^T feeds the BPTT chain.  Where the chain is rec_bwd_x6p, the split-K slabs of the dh
        // GEMM stay unreduced and the chain's prologue adds them: one launch (7 us + its gap) less in front of it
        int keep = 0;
        bool fold = false;
        if (!sg && y.D == 1 && R == y.Bp && !simple_rec(h)) {
            const bool fold_on = h->fold_dh;
            RecArgs ra = rec_args(h, y.L - 1);
            fold = fold_on && sbr_rec_x6p_ok(ra) && !sbr_rec_cluster_ok(ra);
        }
        // Round 5: logits, softmax + CCE and dh in ONE launch whose workgroups exchange the row statistics inside the kernel
        // (sbr_head.hip; exact-f32 products); its dh leaves as split-K slabs -- folded into the chain's prologue as above, or
        // reduced here.  Shapes it does not serve (and SBR_HEAD_FUSE=0) keep the three launches below.
        bool head_done = false;
        if (h->head_fuse && y.cfg.loss == SBR_LOSS_CCE && !sg && !bf16p && !(y.cfg.flags & SBR_FLAG_F32_MFMA) && y.D == 1 && R == y.Bp) {
            int nsl = 0; hipError_t he = hipSuccess;
            h->head_epoch += 1; if (!h->head_epoch) h->head_epoch = 1;
            if (launch_head_cce(s, hl, h->P(y.p_WoutT), h->P(y.p_bout), tgt, h->bpop, lg, h->A(y.a_rowcost), ws, y.ws_floats,
                                (unsigned*)h->A(y.a_hstat), (int*)h->A(y.a_fault), y.Bp, N, Nl, Hp, y.Bg, h->head_epoch, &nsl, &he,
                                (y.cfg.flags & SBR_FLAG_PROFILE_REC) && (size_t)y.Bp * 16 >= 256 * 8 ? (unsigned long long*)h->A(y.a_prof) + (size_t)2 * (y.Bp / 16) * 16 * 8 : nullptr)) {
                SBR_LAUNCH(he);
                head_done = true;
                if (fold) keep = nsl;
                else SBR_LAUNCH(launch_splitk_reduce(s, ws, nsl, y.Bp, Hp, h->A(y.a_dhlast), Hp, nullptr));
            }
        }
        if (!head_done) {
        if (bf16p) sbr_gemm_set_planes(1);
        else if (layer_gemm_f16(h, false)) sbr_gemm_hint(2, 1.0f, 1.0f);      // h in [-1, 1] x weights
        const hipError_t ge = launch_gemm(s, hl, Hp, 1, h->P(y.p_WoutT), 1, Hp, lg, Nl, R, N, Hp, nullptr, nullptr, 0, sg);
        sbr_gemm_set_planes(3);
        SBR_LAUNCH(ge);
        if (SBR_LOSS_IS_MARGIN(y.cfg.loss))
            SBR_LAUNCH(launch_margin_loss(s, lg, h->P(y.p_bout), tgt, y.NT, h->bX, h->blen, y.T, y.F, h->A(y.a_dflt), h->A(y.a_rowcost), R, N, Nl,
                                          y.Bg, y.cfg.loss, y.cfg.balance, y.cfg.unique));
        else
        SBR_LAUNCH(launch_softmax_cce(s, lg, h->P(y.p_bout), tgt, h->bpop, h->A(y.a_rowcost), R, N, Nl, y.Bg));
        SBR_LAUNCH(launch_gemm(s, lg, Nl, 1, h->P(y.p_WoutT), Hp, 1, h->A(y.a_dhlast), Hp, R, Hp, N, nullptr, ws, y.ws_floats, sg, 0, 0,
                               fold ? &keep : nullptr));
        }
        h->dh_slabs_n = keep;
        // beside the BPTT chain: cost, db_out (+ bias regulariser), dW_out^T [N][Hp] = dlogits^T . h.  One record at the end
        // of this phase's main-stream work releases the side stream and is the timing mark in front of rec_bwd.
        h->ev_lg_rec = record_shared(h, h->ev_lg, 3); h->lg_seq = h->batch_seq;
        SBR_HIP(hipStreamWaitEvent(sd, h->ev_lg_rec, 0));
        if (!fill_needed) { const int rc = side_batch_work(); if (rc != SBR_OK) return rc; }
        // Overlapped tail, single-call step: the output layer's gradient kernels and its update (five launches, 50 - 60 us on one
        // stream with its gaps) go to the SECOND side stream, in front of the scatter-add, so that the polling weight-gradient
        // GEMM on `sd` starts with the chain instead of a third of it late (profiles/round3_l_timeline.txt, round3_z_timeline0.txt):
        // the scatter-add's units catch up with what was released meanwhile in one pass, the GEMM's groups would carry the backlog
        // to the end.  (A stream of their own was tried: with main, two side streams and the monitor's that is a fifth hardware
        // queue, and two of them then share one -- profiles/round3_A_variants.txt.)  No split-K workspace for dW_out there: the
        // polling GEMM owns ws2 meanwhile.
        h->out3 = h->in_train_step && h->tail_nc >= 2 && h->tail_overlap == 1 && h->tail_out_stream;
        // SBR_TAIL_OUT_STREAM=2: ... or the monitor's stream, which is idle while the scatter-add launch carries the monitor
        h->out3_stream = (h->tail_out_stream == 2 && h->tail_mon_units && h->tail_cost_scanned) ? h->side3 : h->side2;
        hipStream_t so = h->out3 ? h->out3_stream : sd;
        if (h->out3) SBR_HIP(hipStreamWaitEvent(so, h->ev_lg_rec, 0));
        // Single-call step without a bias regulariser: the output layer's gradient, its step and the batch cost in ONE launch
        // (launch_out_grad_step, sbr_misc.hip) instead of the five or six below -- the polling weight-gradient GEMM of the overlapped
        // tail, next on this stream, then starts with the chain instead of 68 us into it.  SBR_OUT_FUSE=0: as before.
        h->out_stepped = false;
        // (taken WITHOUT the overlapped tail only: in front of the polling GEMM of C2 it measured 0.3334 against 0.3294 ms -- that GEMM
        // then starts 9 us earlier and ends where it did, it is throughput-bound beside the chain; C1: 0.3156 -> 0.3035 together with
        // the one-launch head: profiles/round5_variants.txt call b)
        const bool will_step_here = h->in_train_step && !y.n_sparse && !sg && h->tail_nc == 0;
        if (h->out_fuse && will_step_here && y.cfg.regularization == 0.0f && y.D == 1) {
            hipError_t oe = hipSuccess;
            float* s1e = y.n_state_arrays > 1 ? h->St(1, 0) : nullptr;
            if (launch_out_grad_step(so, lg, hl, h->A(y.a_rowcost), h->cost_ptr(), y.cfg.updater, h->P(y.p_WoutT), h->St(0, y.p_WoutT),
                                     s1e ? s1e + y.p_WoutT : nullptr, h->P(y.p_bout), h->St(0, y.p_bout), s1e ? s1e + y.p_bout : nullptr,
                                     R, N, Nl, Hp, y.cfg.learning_rate, y.cfg.rho, y.cfg.beta1, y.cfg.beta2, (long)h->step_count + 1, &oe)) {
                SBR_LAUNCH(oe);
                h->out_stepped = true;
            }
        }
        if (!h->out_stepped) {
        SBR_LAUNCH(launch_sum_cost(so, h->A(y.a_rowcost), R, h->cost_ptr()));
        // data-parallel: every rank adds its share of the bias regulariser, shares sum to reg
        const float reg = y.cfg.regularization * (float)R / (float)y.Bg;
        SBR_LAUNCH(launch_colsum_bias(so, lg, R, N, Nl, h->Gd(y.p_bout), h->P(y.p_bout), reg, h->cost_ptr(), h->A(y.a_csum)));
        SBR_LAUNCH(launch_gemm(so, lg, 1, Nl, hl, Hp, 1, h->Gd(y.p_WoutT), Hp, N, Hp, R, nullptr, h->out3 ? h->A(y.a_ws3) : ws2,
                               h->out3 ? y.ws3_floats : y.ws2_floats, sg));
        }
        SBR_HIP(hipEventRecord(h->ev_og, so)); h->og_recorded = true;   // output-layer gradients + cost complete
        // Single-call step, dense updates: the output layer is stepped right here, beside the BPTT chain (nothing reads W_out
        // any more: dh was computed in front of the record the side stream waited on); sbr_apply_update leaves the range
        // out.  C4: 46 us off the end of the step.  (The overlapped tail does the same itself; phase-by-phase callers --
        // data parallel -- reduce the gradients first.)
        if (h->in_train_step && !y.n_sparse && h->tail_nc == 0 && !simple_gemm(h)) {
            float* s1e = y.n_state_arrays > 1 ? h->St(1, 0) : nullptr;
            if (!h->out_stepped)
            SBR_LAUNCH(launch_update(sd, y.cfg.updater, h->P(y.p_split), h->Gd(y.p_split), h->St(0, y.p_split), s1e ? s1e + y.p_split : nullptr,
                                     y.n_params - y.p_split, y.cfg.learning_rate, y.cfg.rho, y.cfg.beta1, y.cfg.beta2, (long)h->step_count + 1));
            h->out_early = true;
        }
    } else {
        const int C = y.C;
        int* cells = (int*)h->A(y.a_cells);
        float *Wc = h->A(y.a_Wc), *bc = h->A(y.a_bc), *act = h->A(y.a_act), *dWc = h->A(y.a_dWc), *dbc = h->A(y.a_dbc);
        if (h->cells_early) SBR_HIP(hipStreamWaitEvent(s, h->ev_cells, 0));      // built and caught up beside the forward chain (sbr_forward)
        else {
        SBR_LAUNCH(launch_build_cells(s, tgt, h->bsmp, y.Bg, y.S, cells));
        if (sparse_lazy(h))      // ... and so must the rows of W_out^T / b_out the sampled cells gather
            for (int b = 0; b < y.n_sparse; ++b)
                if (y.sparse[b].kind == 1)
                    SBR_LAUNCH(launch_sparse_catch_up_list(s, sparse_rows(h, b), sparse_upd(h), cells, nullptr, C, C, (int)h->step_count));
        }
        h->cells_early = false;
        SBR_LAUNCH(launch_gather_rows(s, h->P(y.p_WoutT), h->P(y.p_bout), cells, C, Hp, Wc, bc));
        // Round 6: activations, loss, its gradient and dh in ONE launch where the shape allows it (head_sampled_kernel, sbr_head.hip):
        // four launches on twenty workgroups each were 113 us between the two chains of C3.  SBR_HEAD_FUSE=0: the launches below.
        bool head1 = false;
        if (h->head_fuse && !sg && y.D == 1 && !(y.cfg.flags & SBR_FLAG_F32_MFMA)) {
            hipError_t he = hipSuccess;
            head1 = launch_head_sampled(s, hl, Wc, bc, h->bpop, act, h->A(y.a_rowcost), h->A(y.a_dhlast), R, C, Hp, y.Bg, y.S,
                                        y.cfg.row_offset, y.cfg.loss, y.Bg, &he,
                                        (y.cfg.flags & SBR_FLAG_PROFILE_REC) ? (unsigned long long*)h->A(y.a_prof) + (size_t)2 * (y.Bp / 16) * 16 * 8 : nullptr);
            if (head1) SBR_LAUNCH(he);
        }
        if (!head1) {
        SBR_LAUNCH(launch_gemm(s, hl, Hp, 1, Wc, 1, Hp, act, C, R, C, Hp, nullptr, nullptr, 0, sg));
        SBR_LAUNCH(launch_sampled_loss(s, act, bc, h->bpop, h->A(y.a_rowcost), R, y.Bg, y.S, y.cfg.row_offset,
                                       y.cfg.loss, y.Bg));
        }
        // Round 5: dh feeds the BPTT chain, everything else here only feeds the optimizer -- cost sum, bias column sums, the dWc GEMM
        // and the scatter of the cells' gradients (5 launches, ~75 us at C3 beside the side stream's sort) leave the main stream: dh
        // first, one record, the rest on the side stream beside the chain (as the dense heads always did).  SBR_SAMPLED_SIDE=0: rounds 1 - 4.
        const int sampled_side = 1;
        hipStream_t sg_s = sampled_side ? sd : s;
        if (sampled_side) {
            if (!head1) SBR_LAUNCH(launch_gemm(s, act, C, 1, Wc, Hp, 1, h->A(y.a_dhlast), Hp, R, Hp, C, nullptr, nullptr, 0, sg));
            h->ev_lg_rec = record_shared(h, h->ev_lg, 3); h->lg_seq = h->batch_seq;
            SBR_HIP(hipStreamWaitEvent(sd, h->ev_lg_rec, 0));
        }
        SBR_LAUNCH(launch_sum_cost(sg_s, h->A(y.a_rowcost), R, h->cost_ptr()));
        SBR_LAUNCH(launch_colsum_bias(sg_s, act, R, C, C, dbc, nullptr, 0.0f, nullptr, h->A(y.a_csum)));
        SBR_LAUNCH(launch_gemm(sg_s, act, 1, C, hl, Hp, 1, dWc, Hp, C, Hp, R, nullptr, nullptr, 0, sg));
        if (!sampled_side) SBR_LAUNCH(launch_gemm(s, act, C, 1, Wc, Hp, 1, h->A(y.a_dhlast), Hp, R, Hp, C, nullptr, nullptr, 0, sg));
        SBR_LAUNCH(launch_scatter_cells(sg_s, h->Gd(y.p_WoutT), h->Gd(y.p_bout), dWc, dbc, cells, C, Hp));
        SBR_HIP(hipEventRecord(h->ev_og, sg_s)); h->og_recorded = true;
        if (!sampled_side) h->ev_lg_rec = h->ev_og;
        if (!fill_needed) {      // the batch-only side work follows (the side stream has waited for this phase's record)
            if (!sampled_side) SBR_HIP(hipStreamWaitEvent(sd, h->ev_og, 0));
            const int rc = side_batch_work(); if (rc != SBR_OK) return rc;
        }
        // Single-call step: the head's row-sparse block (W_out^T rows + b_out of the sampled cells) has its complete gradient now
        // and nothing reads those rows any more (dh is computed): its step runs on the side stream beside the BPTT chain instead
        // of at the end of the step (C3: 35 us, C5: 41 us); sbr_apply_update leaves the block out.
        h->wout_early = false;
        if (h->in_train_step && h->sparse_out_early)
            for (int b = 0; b < y.n_sparse; ++b)
                if (y.sparse[b].kind == 1 && !h->sp_exchanged[b]) {
                    SBR_HIP(hipStreamWaitEvent(sd, h->ev_og, 0));      // (recorded on this very stream unless SBR_SAMPLED_SIDE=0)
                    SBR_LAUNCH(launch_sparse_step_list(sd, sparse_rows(h, b), sparse_upd(h), (const int*)h->A(y.a_cells), nullptr, C, C,
                                                       (int)h->step_count + 1));
                    h->wout_early = true;
                }
    }
    mark(h, 3);
    if (h->deferred_join && !h->in_train_step) {
        // the caller orders its collective behind the SIDE stream: make that stream also cover what this phase wrote
        // to the output-layer gradients on the main stream (sampled heads)
        SBR_HIP(hipEventRecord(h->ev_lg, s));
        SBR_HIP(hipStreamWaitEvent(sd, h->ev_lg, 0));
        return SBR_OK;
    }
    // called on its own (the caller reads the output-layer gradients next): join now
    if (!h->in_train_step) return side_join(h);
    return SBR_OK;
}

extern "C" int sbr_backward_recurrent(sbr_handle* h) {
    CHECK_ARG(h, "null handle");
    sbr_gemm_set_exact_f32((h->lay.cfg.flags & SBR_FLAG_F32_MFMA) != 0);
    if (!h->fwd_done) { sbr_set_error("sbr_backward_recurrent: call sbr_forward first"); return SBR_ESTATE; }
    const Layout& y = h->lay; hipStream_t s = h->stream, sd = h->side;
    const bool sg = simple_gemm(h);
    h->grads_clean = false;
    if (y.D == 2) return backward_bi(h);
    float* ws = h->A(y.a_ws);
    float* ws2 = h->A(y.a_ws2);
    const int TB = y.T * y.Bp;
    for (int l = y.L - 1; l >= 0; --l) {
        const LayerLayout& ly = y.layer[l];
        const int GHp = y.G * ly.Hp;
        RecArgs a = rec_args(h, l);
        if (h->fill_done && sbr_rec_cluster_ok(a)) {
            a.sentinel_done = 1;
            if (l == y.L - 1) SBR_HIP(hipStreamWaitEvent(s, h->ev_fill, 0));
        }
        a.dh_last = l == y.L - 1 ? h->A(y.a_dhlast) : nullptr;
        if (l == y.L - 1 && h->dh_slabs_n > 0) { a.dh_slabs = ws; a.n_dh_slabs = h->dh_slabs_n; }      // (sbr_loss_backward_output)
        a.dh_ext = l < y.L - 1 ? h->A(ly.a_dhext) : nullptr;
        if (a.prof) a.prof += (size_t)(y.Bp / 16) * 16 * 8;
        const int nblk = sbr_rec_bwd_blocks(a, simple_rec(h));
        // BPTT in time chunks when the bf16x6 kernel runs: dW_hid of a finished chunk is computed on the side
        // stream (190 idle CUs) while the chain continues
        const size_t slab = (size_t)ly.Hp * GHp;
        int nc = (sbr_rec_bwd_chunkable(a, simple_rec(h)) && y.T >= 64 && !sg) ? h->bwd_chunks : 1;
        int nsl = (int)std::min<size_t>(h->wgrad_slices / nc, y.ws2_floats / (slab * nc));   // K-slices (= workgroups of the wgrad kernel)
        if (nsl < 1) nc = 1;
        const bool side_wgrad = !simple_rec(h) && !sg && nsl >= 1;   // weight gradients on the side stream
        // the bf16x6 GEMM covers the slab with 128x128 tiles: ~512 workgroups in all is enough (the dedicated f32
        // kernel, one workgroup per slab, wants many thin slabs)
        const bool wg_gemm = (h->wgrad_x6 && !(y.cfg.flags & SBR_FLAG_F32_MFMA) && ly.Hp >= 96) || !(ly.Hp == 32 || ly.Hp == 64 || ly.Hp == 128);
        // fp16 x3 products for that GEMM: its operands are hidden states (|h| <= 1 behind tanh / sigmoid gates) and gradients
        // that have passed the clip at +-100 (scaled by 2^9 into fp16's range), see gemm_x6_kernel NP = 2
        const int wgf = h->wgrad_f16;
        const bool wg_f16 = wgf && !a.relu && y.cfg.grad_clip > 0.0f && y.cfg.grad_clip <= 100.0f;
        if (wg_gemm && nsl > 1) {
            const int wgs = h->wgrad_x6_wgs;
            nsl = std::max(1, std::min(nsl, wgs / (((ly.Hp + 127) / 128) * ((GHp + 127) / 128)) / nc));
        }
        // Tail of a single-layer step with one BPTT launch: the main stream keeps the longer branch (dW_hid GEMM + slab
        // reduction + its updates) and the side stream takes the bias partials, the embedding scatter-add and their
        // updates -- the main stream then ends the step without waiting ~13 us for a cross-stream event behind the
        // branch that finishes last (profiles/round1_i_timeline.txt).
        const bool swap = h->swap_tail && side_wgrad && nc == 1 && y.L == 1 && !y.E && !y.n_sparse &&
                          y.n_params <= ((size_t)4 << 20) &&      // large models (C4: 34 M parameters) measured 2 % slower this way
                          !(y.cfg.flags & SBR_FLAG_ATOMIC_SCATTER);   // phase-by-phase callers (data parallel) join the side stream
                                                                      // before their collective: same split of the tail
        hipStream_t sw = swap ? s : sd;      // weight-gradient GEMM
        hipStream_t sm = swap ? sd : s;      // partials + scatter
        if (l == 0 && y.L == 1 && h->tail_nc >= 2) {
            // ---- Overlapped tail.  The chain (64 of 256 CUs at C2) stores dxt / dhi write-through and every wave publishes the
            // time step it has completed.  Two consumers run beside it on the idle CUs, ONE launch each, whose workgroups /
            // waves wait inside the kernel for the time steps they read (SbrPoll, sbr_common.h):
            //   side stream   (the output layer's gradient kernels, left over from the loss phase ->) gate (returns once every wave
            //                 of the chain has published: the chain is resident, spinning consumers can no longer keep it off the
            //                 chip) -> dW_hid GEMM: persistent groups of workgroups share the K slabs of a table in the order the
            //                 chain releases them, one partial each -> reduction of the partials (-> W_hid update)
            //   second side   (the time-chunked sort and the ids' running cost, beside the FORWARD chain ->) the same gate ->
            //   stream        embedding scatter-add: units that own id ranges of equal cost add their rows in LDS and store each
            //                 once; workgroup 0 of that launch is the MONITOR, which folds the chain's progress words into the
            //                 word every consumer polls (-> W_in update)
            //   main stream   chain -> bias / init-state partial sums (-> their update) -> joins both
            // In a single-call step every stream applies the optimizer to what it has produced (the output layer early, on
            // the side stream); phase-by-phase callers (data parallel) get complete gradients and update in sbr_apply_update.
            const int tnc = h->tail_nc, CH = h->tail_ch;
            // SBR_TAIL_OVERLAP=2: the same kernels, all on the main stream behind the chain (nothing has to run concurrently):
            // for tools that serialise kernels (rocprofv3 --pmc) and for triage
            const bool serial = h->tail_overlap == 2;
            hipStream_t s2 = serial ? s : h->side2;
            if (serial) {
                sd = s;
                { const int rc = side_join(h); if (rc != SBR_OK) return rc; }
                SBR_HIP(hipEventRecord(h->ev_tail2, h->side2));           // the sort ran there
                SBR_HIP(hipStreamWaitEvent(s, h->ev_tail2, 0));
            }
            const bool gru = y.cfg.cell == SBR_CELL_GRU;
            int* words = (int*)h->A(y.a_prog);
            const int nwaves = (y.Bp / a.rpt) * 8;
            int* done = (int*)h->A(y.a_done);
            h->prog_epoch = (h->prog_epoch + 1) & 0x7FFFF; if (!h->prog_epoch) h->prog_epoch = 1;
            a.progress = words; a.prog_every = h->tail_pub_every; a.prog_epoch = h->prog_epoch;
            const int K = y.T * y.Bp;
            const int cap = (int)std::min<size_t>(256, y.ws2_floats / slab);
            SbrPoll pl{words, nwaves, done, a.prog_epoch, y.Bp, a.fault, 0, 0, h->tail_trace, nullptr, 0};
            if (h->tail_slab_key[0] != K || h->tail_slab_key[1] != cap || !h->tail_slab_dev) {      // (first step of this shape)
                sbr_tail_slab_table(K, y.Bp, 255, h->tail_slab_growth, h->tail_slab_max, h->tail_slab_host);
                if (!h->tail_slab_dev) SBR_HIP(hipMalloc(&h->tail_slab_dev, 260 * sizeof(int)));
                SBR_HIP(hipMemcpy(h->tail_slab_dev, h->tail_slab_host.data(), h->tail_slab_host.size() * sizeof(int), hipMemcpyHostToDevice));
                h->tail_slab_key[0] = K; h->tail_slab_key[1] = cap;
            }
            pl.slab_lo = h->tail_slab_dev;
            pl.n_slabs = (int)h->tail_slab_host.size() - 1;
            const int n_slabs = std::max(1, std::min(std::min(h->tail_gemm_groups, cap), pl.n_slabs));      // partials = persistent groups
            const bool upd_here = h->in_train_step;
            float* s1a = y.n_state_arrays > 1 ? h->St(1, 0) : nullptr;
            auto upd_on = [&](hipStream_t st, size_t lo, size_t hi, size_t gap_at = (size_t)-1, size_t gap_len = 0) -> hipError_t {
                return launch_update(st, y.cfg.updater, h->P(lo), h->Gd(lo), h->St(0, lo), s1a ? s1a + lo : nullptr, hi - lo - gap_len,
                                     y.cfg.learning_rate, y.cfg.rho, y.cfg.beta1, y.cfg.beta2, (long)h->step_count + 1, gap_at, gap_len);
            };
            if (!serial) a.fence_kb = h->tail_fence_kb;              // the chain's CUs are its own: the consumers take the other 192
            SBR_LAUNCH_CHAIN(1, s, launch_rec_backward(s, a, false));
            mark(h, 4);
            // side stream: output layer first (its gradients are complete on this stream: dW_out GEMM, bias sums)
            const bool out_early = upd_here && (y.cfg.loss == SBR_LOSS_CCE || SBR_LOSS_IS_MARGIN(y.cfg.loss));
            if (out_early) {
                if (!h->out_stepped)       // (else: launch_out_grad_step has stepped the output layer with its gradient, sbr_loss_backward_output)
                SBR_LAUNCH(upd_on(h->out3 ? h->out3_stream : sd, y.p_split, y.n_params));
                if (h->out3) SBR_HIP(hipEventRecord(h->ev_tail3, h->out3_stream));
            }
            // the monitor: on a stream of its own behind nothing but the chain's first progress words (its own loop waits for them)
            // ... unless the scatter-add launch carries it (default where that launch is the LDS-row one and has its own stream)
            const bool mon_in_units = !serial && h->tail_mon_units && h->tail_cost_scanned;
            if (!mon_in_units) SBR_LAUNCH(launch_tail_monitor(serial ? s : h->side3, pl, a.t_lo));
            SBR_LAUNCH(launch_tail_gate(sd, words, nwaves, a.prog_epoch, y.T, a.fault));
            {
                hipError_t we = hipSuccess;
                if (!launch_gemm_slabs_x6_poll(sd, h->A(ly.a_hs), 1, ly.Hp, a.dxt, GHp, 1, ly.Hp, GHp, K, ws2, n_slabs, GHp, slab,
                                               gru ? a.dhi : nullptr, ly.Hp, gru ? 2 * ly.Hp : 0, &we, wg_f16 ? 2 : 3, 1.0f, wg_f16 ? 512.0f : 1.0f, pl)) {
                    sbr_set_error("overlapped tail: the weight-gradient GEMM rejected the shape"); return SBR_EINVAL;
                }
                SBR_LAUNCH(we);
            }
            if (!serial) SBR_LAUNCH(launch_tail_gate(s2, words, nwaves, a.prog_epoch, y.T, a.fault));
            // (tried and dropped: the last time chunk as a launch of its own behind the polling one, one wave per 16 entries on the
            // then idle chip -- the hot rows' atomics serialise there: 23 us for 6400 entries, profiles/round3_variants.txt call d)
            hipError_t se = hipSuccess;
            if (h->tail_cost_scanned && launch_scatter_lds_poll(s2, h->Gd(ly.p_Win), a.dxt, (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos),
                                                               (const int*)h->A(y.a_soff), (const int*)h->A(y.a_sP), y.cfg.input_size, tnc,
                                                               y.T * y.Bp * y.F, GHp, pl, h->tail_bounds, h->tail_scatter_units, &se,
                                                               mon_in_units, a.t_lo)) {
                SBR_LAUNCH(se);
            } else
            SBR_LAUNCH(launch_scatter_reduce_poll(s2, h->Gd(ly.p_Win), a.dxt, (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos),
                                                  (const int*)h->A(y.a_soff), y.cfg.input_size, tnc, CH, y.T * y.Bp * y.F, GHp, y.Bp, pl,
                                                  0, &h->tail_bounds, h->tail_short_chunks, !serial && h->tail_fence_kb > 0));
            if (upd_here) SBR_LAUNCH(upd_on(s2, ly.p_Win, ly.p_b));
            SBR_HIP(hipEventRecord(h->ev_tail2, s2));
            // single-call step: the slab reduction IS the W_hid update (one launch, one pass less behind the chain); phase-by-phase
            // callers (data parallel) need the reduced gradient
            const int fuse_slabs = h->tail_fuse_slabs;
            if (upd_here && fuse_slabs && ly.p_peep - ly.p_Whid == slab && (slab & 3) == 0) {
                SBR_LAUNCH(launch_update_from_slabs(sd, y.cfg.updater, ws2, n_slabs, h->P(ly.p_Whid), h->St(0, ly.p_Whid),
                                                    s1a ? s1a + ly.p_Whid : nullptr, slab, y.cfg.learning_rate, y.cfg.rho, y.cfg.beta1,
                                                    y.cfg.beta2, (long)h->step_count + 1));
            } else {
                SBR_LAUNCH(launch_splitk_reduce(sd, ws2, n_slabs, ly.Hp, GHp, h->Gd(ly.p_Whid), GHp, nullptr));
                if (upd_here) SBR_LAUNCH(upd_on(sd, ly.p_Whid, ly.p_peep));
            }
            SBR_HIP(hipEventRecord(h->ev_tail, sd));
            // main stream, behind the chain
            SBR_LAUNCH(launch_rec_reduce_partials(s, a.part, nblk, y.G, ly.Hp, y.cfg.cell, h->Gd(ly.p_b), h->Gd(ly.p_peep),
                                                  h->Gd(ly.p_cinit), h->Gd(ly.p_hinit)));
            mark(h, 5);
            if (upd_here) {      // b, then (behind the gap that is W_hid) peepholes / initial states, and the output layer unless done
                // (a sampled head's gradient kernels run on the side stream since round 5 -- SBR_SAMPLED_SIDE -- and this launch reads
                // and clears their output: order it behind them.  Without the wait the chain's length hid the race.)
                if (!out_early && h->og_recorded) SBR_HIP(hipStreamWaitEvent(s, h->ev_og, 0));
                SBR_LAUNCH(upd_on(s, ly.p_b, out_early ? y.p_split : y.n_params, ly.p_Whid - ly.p_b, ly.p_peep - ly.p_Whid));
                h->tail_updated = true;
            }
            if (h->deferred_join && !h->in_train_step && !serial) {
                // data-parallel driver: it orders one collective behind each producing stream (W_in: second side stream,
                // W_hid: side stream, the rest: this stream) and joins through sbr_join_side / sbr_apply_update
                h->tail_join_pending = true;
                mark(h, 6);
                continue;
            }
            SBR_HIP(hipStreamWaitEvent(s, h->ev_tail2, 0));
            mark(h, 6);
            SBR_HIP(hipStreamWaitEvent(s, h->ev_tail, 0));
            if (h->out3 && out_early) SBR_HIP(hipStreamWaitEvent(s, h->ev_tail3, 0));
            else if (h->out3) SBR_HIP(hipStreamWaitEvent(s, h->ev_og, 0));
            h->out3 = false;
            h->side_pending = false;
            continue;
        }
        hipEvent_t ev_chain_end = nullptr;      // this layer: the main-stream record behind its (last) BPTT launch
        if (nc > 1 || side_wgrad) {
            for (int c = 0; c < nc; ++c) {
                a.t_hi = (int)((long)y.T * (nc - c) / nc); a.t_lo = (int)((long)y.T * (nc - c - 1) / nc); a.chunk = c;
                SBR_LAUNCH_CHAIN(1, s, launch_rec_backward(s, a, false));
                ev_chain_end = record_shared(h, h->ev_chunk[c], (l == 0 && c == nc - 1) ? 4 : -1);
                SBR_HIP(hipStreamWaitEvent(sd, ev_chain_end, 0));
                // dW_hid [Hp][G*Hp] += hs[t]^T . dhi[t] over the chunk's positions (hs slot t = h_{t-1})
                const float* hsc = h->A(ly.a_hs) + (size_t)a.t_lo * y.Bp * ly.Hp;
                const int Kc = (a.t_hi - a.t_lo) * y.Bp;
                float* slabs = ws2 + (size_t)c * nsl * slab;
                const bool gru = y.cfg.cell == SBR_CELL_GRU;
                const float* dxc = a.dxt + (size_t)a.t_lo * y.Bp * GHp;
                const float* dhcc = gru ? a.dhi + (size_t)a.t_lo * y.Bp * ly.Hp : nullptr;
                hipError_t we = hipSuccess;
                // swapped tail: this GEMM's slabs share the second workspace with the split-K slabs of the side stream's dW_out
                // GEMM -- normally long reduced by now, but nothing ordered the two (a wait on a complete event is free)
                if (sw == s && h->og_recorded) SBR_HIP(hipStreamWaitEvent(s, h->ev_og, 0));
                if (!wg_gemm && launch_wgrad_slabs(sw, hsc, dxc, dhcc, slabs, ly.Hp, GHp, Kc, nsl, &we)) {
                    SBR_LAUNCH(we);
                } else if (launch_gemm_slabs_x6(sw, hsc, 1, ly.Hp, dxc, GHp, 1, ly.Hp, GHp, Kc, slabs, nsl, GHp, slab, dhcc, ly.Hp,
                                                gru ? 2 * ly.Hp : 0, &we, wg_f16 ? 2 : 3, 1.0f, wg_f16 ? 512.0f : 1.0f)) {
                    SBR_LAUNCH(we);     // one bf16x6 GEMM: columns [0, 2Hp) from dxt, the candidate-gate columns from the compact array
                } else if (gru) {   // hid_input grad = [dxt_r | dxt_u | dhi_c]
                    SBR_LAUNCH(launch_gemm_slabs(sw, hsc, 1, ly.Hp, dxc, GHp, 1, ly.Hp, 2 * ly.Hp, Kc, slabs, nsl, GHp, slab));
                    SBR_LAUNCH(launch_gemm_slabs(sw, hsc, 1, ly.Hp, dhcc, ly.Hp, 1, ly.Hp, ly.Hp, Kc, slabs + 2 * ly.Hp, nsl, GHp, slab));
                } else {
                    SBR_LAUNCH(launch_gemm_slabs(sw, hsc, 1, ly.Hp, dxc, GHp, 1, ly.Hp, GHp, Kc, slabs, nsl, GHp, slab));
                }
            }
            SBR_LAUNCH(launch_splitk_reduce(sw, ws2, nc * nsl, ly.Hp, GHp, h->Gd(ly.p_Whid), GHp, nullptr));
            h->side_pending = true;
            if (l == 0) mark(h, 4);
            if (swap) h->tail_swapped = true;
            SBR_LAUNCH(launch_rec_reduce_partials(sm, a.part, nc * nblk, y.G, ly.Hp, y.cfg.cell, h->Gd(ly.p_b), h->Gd(ly.p_peep),
                                                  h->Gd(ly.p_cinit), h->Gd(ly.p_hinit)));
        } else {
            SBR_LAUNCH_CHAIN(1, s, launch_rec_backward(s, a, simple_rec(h)));
            if (l == 0) mark(h, 4);
            SBR_LAUNCH(launch_rec_reduce_partials(s, a.part, nblk, y.G, ly.Hp, y.cfg.cell, h->Gd(ly.p_b), h->Gd(ly.p_peep),
                                                  h->Gd(ly.p_cinit), h->Gd(ly.p_hinit)));
            if (y.cfg.cell == SBR_CELL_GRU) {
                SBR_LAUNCH(launch_gemm(s, h->A(ly.a_hs), 1, ly.Hp, a.dxt, GHp, 1, h->Gd(ly.p_Whid), GHp, ly.Hp, 2 * ly.Hp, TB, nullptr,
                                       ws, y.ws_floats, sg));
                SBR_LAUNCH(launch_gemm(s, h->A(ly.a_hs), 1, ly.Hp, a.dhi, ly.Hp, 1, h->Gd(ly.p_Whid) + 2 * ly.Hp, GHp, ly.Hp, ly.Hp, TB,
                                       nullptr, ws, y.ws_floats, sg));
            } else {
                SBR_LAUNCH(launch_gemm(s, h->A(ly.a_hs), 1, ly.Hp, a.dxt, GHp, 1, h->Gd(ly.p_Whid), GHp, ly.Hp, GHp, TB, nullptr, ws,
                                       y.ws_floats, sg));
            }
        }
        if (l == 0 && y.E) {
            mark(h, 5);
            // dense layer 0 behind the embedding: dW_in = emb^T . dxt, d_emb = dxt . W_in^T, then the scatter-add of the
            // F*T*B embedding-gradient rows into dW_emb (EmbeddingLayer gradient: duplicates accumulate [3P])
            SBR_LAUNCH(launch_gemm(s, h->A(y.a_emb), 1, ly.n_in_p, a.dxt, GHp, 1, h->Gd(ly.p_Win), GHp, ly.n_in_p, GHp, TB, nullptr, ws,
                                   y.ws_floats, sg));
            SBR_LAUNCH(launch_gemm(s, a.dxt, GHp, 1, h->P(ly.p_Win), 1, GHp, h->A(y.a_demb), ly.n_in_p, TB, ly.n_in_p, GHp, nullptr,
                                   nullptr, 0, sg));
            SBR_HIP(hipStreamWaitEvent(s, h->ev_sort, 0));
            SBR_LAUNCH(launch_scatter_reduce(s, h->Gd(y.p_Emb), h->A(y.a_demb), (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos),
                                             (const int*)h->A(y.a_soff), y.cfg.input_size, y.T * y.Bp * y.F, y.Ep, y.Bp));
            mark(h, 6);
        } else         if (l == 0) {
            mark_on(h, 5, sm);
            if (y.cfg.flags & SBR_FLAG_ATOMIC_SCATTER) {
                SBR_LAUNCH(launch_scatter_rows(s, h->Gd(ly.p_Win), a.dxt, h->bX, a.len, y.T, y.Bp, y.F, GHp));
            } else {
                if (sm == s) SBR_HIP(hipStreamWaitEvent(s, h->ev_sort, 0));      // (the sort ran on the side stream)
                hipError_t se = hipSuccess;
                static const int range_on = [] { const char* e = getenv("SBR_SCAT_RANGE"); return e ? atoi(e) : 1; }();
                // SBR_SCAT_RANGE: 1 (default) = the range form up to 1024-float rows, the atomic kernel beyond (C5: measured 8.35 against
                // 8.44 - 8.48 ms with either new form); 2 = the segment-parallel form; 0 = the atomic kernel everywhere
                if (range_on == 1 && y.a_srpart && GHp <= 1024 &&
                    launch_scatter_range(sm, h->Gd(ly.p_Win), a.dxt, (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos), (const int*)h->A(y.a_soff),
                                         y.cfg.input_size, GHp, h->A(y.a_srpart), (int*)h->A(y.a_srid), SBR_SCAT_RANGES, &se)) {
                    SBR_LAUNCH(se);
                } else
                if (range_on == 2 && y.a_srpart && launch_scatter_wide(sm, h->Gd(ly.p_Win), a.dxt, (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos),
                                                                  (const int*)h->A(y.a_soff), y.cfg.input_size, y.T * y.Bp * y.F, GHp,
                                                                  h->A(y.a_srpart), (int*)h->A(y.a_srid), y.sr_slots, &se)) {
                    SBR_LAUNCH(se);
                } else
                SBR_LAUNCH(launch_scatter_reduce(sm, h->Gd(ly.p_Win), a.dxt, (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos),
                                                 (const int*)h->A(y.a_soff), y.cfg.input_size, y.T * y.Bp * y.F, GHp, y.Bp));
            }
            mark_on(h, 6, sm);
        } else {
            const LayerLayout& lo = y.layer[l - 1];
            const float* xin = h->A(lo.a_hs) + (size_t)y.Bp * lo.Hp;     // input at step t = h^{l-1}_t = slot t+1
            layer_gemm_hint(h, false, true);                     // dW_in = h^{l-1 T} . dxt
            SBR_LAUNCH(launch_gemm(s, xin, 1, lo.Hp, a.dxt, GHp, 1, h->Gd(ly.p_Win), GHp, lo.Hp, GHp, TB, nullptr, ws,
                                   y.ws_floats, sg));
            layer_gemm_hint(h, true, false);                     // dh^{l-1} = dxt . W_in^T
            SBR_LAUNCH(launch_gemm(s, a.dxt, GHp, 1, h->P(ly.p_Win), 1, GHp, h->A(lo.a_dhext), lo.Hp, TB, lo.Hp, GHp, nullptr,
                                   nullptr, 0, sg));
        }
    }
    if (!h->in_train_step && !h->deferred_join) return side_join(h);
    return SBR_OK;
}

extern "C" int sbr_set_deferred_join(sbr_handle* h, int on) {
    CHECK_ARG(h, "null handle");
    h->deferred_join = on != 0;
    return SBR_OK;
}
extern "C" int sbr_join_side(sbr_handle* h) {
    CHECK_ARG(h, "null handle");
    return side_join(h);
}

// ---------------------------------------------------------------------------------------
// data-parallel exchange of the row-sparse blocks (include/sbr_rnn.h)
// ---------------------------------------------------------------------------------------
extern "C" int sbr_sparse_info(sbr_handle* h, int b, int64_t* n_rows, int64_t* row_floats, int64_t* max_local_rows) {
    CHECK_ARG(h, "null handle");
    CHECK_ARG(b >= 0 && b < h->lay.n_sparse, "sparse block %d outside [0,%d)", b, h->lay.n_sparse);
    const SparseBlockLayout& sb = h->lay.sparse[b];
    if (n_rows) *n_rows = sb.n_rows;
    if (row_floats) *row_floats = sb.W;
    if (max_local_rows) *max_local_rows = sb.max_local;
    return SBR_OK;
}

extern "C" int sbr_sparse_pack(sbr_handle* h, int b, int32_t* ids_dev, float* rows_dev, int32_t* count_host) {
    CHECK_ARG(h && ids_dev && rows_dev && count_host, "null argument");
    CHECK_ARG(b >= 0 && b < h->lay.n_sparse, "sparse block %d outside [0,%d)", b, h->lay.n_sparse);
    const Layout& y = h->lay; const SparseBlockLayout& sb = y.sparse[b];
    { const int rc = side_join(h); if (rc != SBR_OK) return rc; }      // the block's gradients come from both streams
    int* count = (int*)h->A(sb.a_count);
    const int epoch = ++h->sp_epoch;
    if (sb.kind == 0) {
        SBR_LAUNCH(launch_sparse_pack(h->stream, sparse_rows(h, b), (const int*)h->A(y.a_sid), (const int*)h->A(y.a_soff) + y.cfg.input_size, 0,
                                      y.T * y.Bp * y.F, (int*)h->A(sb.a_mark), epoch, ids_dev, rows_dev, sb.W, count));
    } else {
        SBR_LAUNCH(launch_sparse_pack(h->stream, sparse_rows(h, b), (const int*)h->A(y.a_cells), nullptr, y.C, y.C, (int*)h->A(sb.a_mark), epoch,
                                      ids_dev, rows_dev, sb.W, count));
    }
    SBR_HIP(hipMemcpyAsync(count_host, count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SBR_HIP(hipStreamSynchronize(h->stream));
    h->sp_exchanged[b] = 1; h->sp_ncand[b] = 0;
    return SBR_OK;
}

extern "C" int sbr_sparse_unpack_add(sbr_handle* h, int b, const int32_t* ids_dev, const float* rows_dev, int count) {
    CHECK_ARG(h && (count == 0 || (ids_dev && rows_dev)), "null argument");
    CHECK_ARG(b >= 0 && b < h->lay.n_sparse, "sparse block %d outside [0,%d)", b, h->lay.n_sparse);
    const SparseBlockLayout& sb = h->lay.sparse[b];
    if (!h->sp_exchanged[b]) { sbr_set_error("sbr_sparse_unpack_add: call sbr_sparse_pack for this step first"); return SBR_ESTATE; }
    CHECK_ARG(count >= 0 && h->sp_ncand[b] + count <= sb.cand_cap, "%d + %d rows exceed the candidate capacity %d", h->sp_ncand[b], count, sb.cand_cap);
    SBR_LAUNCH(launch_sparse_unpack_add(h->stream, sparse_rows(h, b), ids_dev, rows_dev, count, sb.W, (int*)h->A(sb.a_cand) + h->sp_ncand[b]));
    h->sp_ncand[b] += count;
    h->grads_clean = false;
    return SBR_OK;
}

extern "C" int sbr_sparse_pack_device(sbr_handle* h, int b, int32_t* ids_dev, float* rows_dev) {
    CHECK_ARG(h && ids_dev && rows_dev, "null argument");
    CHECK_ARG(b >= 0 && b < h->lay.n_sparse, "sparse block %d outside [0,%d)", b, h->lay.n_sparse);
    const Layout& y = h->lay; const SparseBlockLayout& sb = y.sparse[b];
    { const int rc = side_join(h); if (rc != SBR_OK) return rc; }      // the block's gradients come from both streams
    const int epoch = ++h->sp_epoch;
    // the running count of the pack kernel IS the in-band word ids_dev[0]
    if (sb.kind == 0) {
        SBR_LAUNCH(launch_sparse_pack(h->stream, sparse_rows(h, b), (const int*)h->A(y.a_sid), (const int*)h->A(y.a_soff) + y.cfg.input_size, 0,
                                      y.T * y.Bp * y.F, (int*)h->A(sb.a_mark), epoch, ids_dev + 1, rows_dev, sb.W, ids_dev));
    } else {
        SBR_LAUNCH(launch_sparse_pack(h->stream, sparse_rows(h, b), (const int*)h->A(y.a_cells), nullptr, y.C, y.C, (int*)h->A(sb.a_mark), epoch,
                                      ids_dev + 1, rows_dev, sb.W, ids_dev));
    }
    h->sp_exchanged[b] = 1; h->sp_ncand[b] = 0;
    return SBR_OK;
}

extern "C" int sbr_sparse_unpack_add_all(sbr_handle* h, int b, const int32_t* ids_all, const float* rows_all, int world) {
    CHECK_ARG(h && ids_all && rows_all, "null argument");
    CHECK_ARG(b >= 0 && b < h->lay.n_sparse, "sparse block %d outside [0,%d)", b, h->lay.n_sparse);
    const SparseBlockLayout& sb = h->lay.sparse[b];
    if (!h->sp_exchanged[b]) { sbr_set_error("sbr_sparse_unpack_add_all: call sbr_sparse_pack_device for this step first"); return SBR_ESTATE; }
    const int cap = sb.max_local;
    CHECK_ARG(world >= 1 && h->sp_ncand[b] == 0 && (long)world * cap <= sb.cand_cap, "%d ranks x %d rows exceed the candidate capacity %d", world, cap, sb.cand_cap);
    for (int r = 0; r < world; ++r)      // rank order, one stream: every replica adds in the same order
        SBR_LAUNCH(launch_sparse_unpack_add_dev(h->stream, sparse_rows(h, b), ids_all + (size_t)r * (cap + 1), rows_all + (size_t)r * cap * sb.W,
                                                cap, sb.W, (int*)h->A(sb.a_cand) + (size_t)r * cap));
    h->sp_ncand[b] = world * cap;          // slots beyond a rank's count hold -1 (skipped by the row-sparse step)
    h->grads_clean = false;
    return SBR_OK;
}

extern "C" int sbr_dense_ranges(sbr_handle* h, int cap, int64_t* lo, int64_t* hi, int* n) {
    CHECK_ARG(h && lo && hi && n, "null argument");
    const Layout& y = h->lay;
    std::vector<std::pair<size_t, size_t>> skip;
    for (int b = 0; b < y.n_sparse; ++b)
        for (int k = 0; k < y.sparse[b].npairs; ++k)
            skip.push_back({y.sparse[b].off[k], y.sparse[b].off[k] + (size_t)y.sparse[b].n_rows * y.sparse[b].stride[k]});
    std::sort(skip.begin(), skip.end());
    std::vector<std::pair<size_t, size_t>> out;
    size_t pos = 0;
    // ranges never straddle the output-layer split: the part in front of it is complete later than the part behind it
    auto emit = [&](size_t a, size_t b) {
        if (b <= a) return;
        if (a < y.p_split && b > y.p_split) { out.push_back({a, y.p_split}); out.push_back({y.p_split, b}); }
        else out.push_back({a, b});
    };
    for (auto& r : skip) { emit(pos, r.first); pos = r.second; }
    emit(pos, y.n_params + 1);                        // the batch cost rides behind the last parameter
    CHECK_ARG((int)out.size() <= cap, "%d ranges, capacity %d", (int)out.size(), cap);
    for (size_t i = 0; i < out.size(); ++i) { lo[i] = (int64_t)out[i].first; hi[i] = (int64_t)out[i].second; }
    *n = (int)out.size();
    return SBR_OK;
}

extern "C" int sbr_apply_update(sbr_handle* h) {
    CHECK_ARG(h, "null handle");
    const Layout& y = h->lay;
    h->step_count += 1;
    if (h->tail_join_pending) { const int rc = side_join(h); if (rc != SBR_OK) return rc; }
    float* s1 = y.n_state_arrays > 1 ? h->St(1, 0) : nullptr;
    auto upd = [&](size_t lo, size_t hi) -> hipError_t {
        if (hi <= lo) return hipSuccess;
        return launch_update(h->stream, y.cfg.updater, h->P(lo), h->Gd(lo), h->St(0, lo), s1 ? s1 + lo : nullptr, hi - lo,
                             y.cfg.learning_rate, y.cfg.rho, y.cfg.beta1, y.cfg.beta2, (long)h->step_count);
    };
    const size_t p_end = h->out_early ? y.p_split : y.n_params;     // the output layer was stepped beside the BPTT chain
    // [0, hi) of the parameter section.
    // Single-call step, dense wide index-input block, the step's plain-key sort at hand (a_soff: this batch's segment offsets): the pass
    // over W_in reads / clears the gradient of the touched rows only (launch_update_rows_aware).  SBR_ROW_AWARE_UPDATE=0: update_kernel.
    const bool row_aware = h->row_aware && h->in_train_step && y.a_srpart && !y.n_sparse && !y.E && y.D == 1 && h->tail_nc < 2 &&
                           !(y.cfg.flags & SBR_FLAG_ATOMIC_SCATTER) && !simple_gemm(h) && !simple_rec(h) && ((y.G * y.layer[0].Hp) & 3) == 0;
    auto upd_front = [&](size_t hi) -> hipError_t {
        if (!row_aware) return upd(0, hi);
        const LayerLayout& l0 = y.layer[0];
        const int GHp0 = y.G * l0.Hp;
        const size_t w_end = l0.p_Win + (size_t)y.cfg.input_size * GHp0;
        hipError_t e = upd(0, l0.p_Win);
        if (e != hipSuccess) return e;
        if (hi < w_end) return hipErrorInvalidValue;      // (callers pass ranges that cover the block)
        e = launch_update_rows_aware(h->stream, y.cfg.updater, h->P(l0.p_Win), h->Gd(l0.p_Win), h->St(0, l0.p_Win), s1 ? s1 + l0.p_Win : nullptr,
                                     y.cfg.input_size, GHp0, (const int*)h->A(y.a_soff), y.cfg.learning_rate, y.cfg.rho, y.cfg.beta1,
                                     y.cfg.beta2, (long)h->step_count);
        if (e != hipSuccess) return e;
        return hi > w_end ? upd(w_end, hi) : hipSuccess;
    };
    if (y.n_sparse) {
        // dense pass over everything outside the sparse blocks, then one row-sparse step per block over the rows this step
        // touched: the scatter's sorted ids / the sampled cells, or (data parallel) the ids gathered from every rank
        // Single rank: the index-input block's row step goes FIRST, in front of the join -- its gradient rows are this stream's own
        // work (the scatter-add), so the pass (HBM-bound, 112 us at C3) runs beside the weight-gradient GEMM the side stream is still
        // busy with instead of behind it (round 6: C3's tail behind the chain 306 -> ~265 us)
        bool stepped[2] = {false, false};
        for (int b = 0; b < y.n_sparse; ++b) {
            const SparseBlockLayout& sb = y.sparse[b];
            if (sb.kind != 0 || h->sp_exchanged[b] || !h->side_pending || y.D != 1 || y.E) continue;      // (plain index input, one direction: the scatter-add ran on this stream)
            const int nmax = y.T * y.Bp * y.F;
            SBR_LAUNCH(launch_sparse_step_list(h->stream, sparse_rows(h, b), sparse_upd(h), (const int*)h->A(y.a_sid),
                                               (const int*)h->A(y.a_soff) + y.cfg.input_size, 0, nmax, (int)h->step_count));
            stepped[b] = true;
        }
        { const int rc = side_join(h); if (rc != SBR_OK) return rc; }
        std::vector<std::pair<size_t, size_t>> skip;
        for (int b = 0; b < y.n_sparse; ++b)
            for (int k = 0; k < y.sparse[b].npairs; ++k)
                skip.push_back({y.sparse[b].off[k], y.sparse[b].off[k] + (size_t)y.sparse[b].n_rows * y.sparse[b].stride[k]});
        std::sort(skip.begin(), skip.end());
        size_t pos = 0;
        for (auto& r : skip) { SBR_LAUNCH(upd(pos, r.first)); pos = r.second; }
        SBR_LAUNCH(upd(pos, y.n_params));
        for (int b = 0; b < y.n_sparse; ++b) {
            const SparseBlockLayout& sb = y.sparse[b];
            const SbrSparseRows rows = sparse_rows(h, b);
            if ((sb.kind == 1 && h->wout_early) || stepped[b]) {
                // stepped already: beside the BPTT chain (sbr_loss_backward_output) / in front of the join above
            } else
            if (h->sp_exchanged[b]) {
                SBR_LAUNCH(launch_sparse_step_list(h->stream, rows, sparse_upd(h), (const int*)h->A(sb.a_cand), nullptr, h->sp_ncand[b],
                                                   h->sp_ncand[b], (int)h->step_count));
            } else if (sb.kind == 0) {
                const int nmax = y.T * y.Bp * y.F;
                SBR_LAUNCH(launch_sparse_step_list(h->stream, rows, sparse_upd(h), (const int*)h->A(y.a_sid),
                                                   (const int*)h->A(y.a_soff) + y.cfg.input_size, 0, nmax, (int)h->step_count));
            } else {
                SBR_LAUNCH(launch_sparse_step_list(h->stream, rows, sparse_upd(h), (const int*)h->A(y.a_cells), nullptr, y.C, y.C,
                                                   (int)h->step_count));
            }
            h->sp_exchanged[b] = 0; h->sp_ncand[b] = 0;
        }
    } else if (h->tail_updated) {
        // overlapped tail of a single-call step: every stream has stepped what it produced (sbr_backward_recurrent)
    } else if (h->side_pending && h->tail_swapped) {
        // side stream: everything but W_hid (W_in, b from its own scatter / partials; the output layer's gradients are its
        // own too); main stream: W_hid (its own GEMM) -- no event wait in front of it; then the main stream joins the side
        // stream, normally done by then
        const LayerLayout& l0 = y.layer[0];
        SBR_LAUNCH(launch_update(h->side, y.cfg.updater, h->P(0), h->Gd(0), h->St(0, 0), s1, l0.p_Whid + (p_end - l0.p_peep),
                                 y.cfg.learning_rate, y.cfg.rho, y.cfg.beta1, y.cfg.beta2, (long)h->step_count, l0.p_Whid,
                                 l0.p_peep - l0.p_Whid));
        SBR_LAUNCH(upd(l0.p_Whid, l0.p_peep));
        { const int rc = side_join(h); if (rc != SBR_OK) return rc; }
    } else if (h->side_pending && h->og_recorded) {
        // The last thing the side stream produces is dW_hid (weight-gradient GEMM + slab reduction, 240 us at C4).  Every
        // other parameter is updated while it finishes: the main stream waits only for the output-layer gradients
        // (recorded long ago), updates all ranges except the W_hid blocks, joins, then updates those.
        if (y.L * y.D == 1 && y.n_params - y.layer[0].p_peep <= ((size_t)1 << 20)) {
            // small output layer (C2: 0.47 M floats): two launches instead of three; a large one (C4: 6.8 M) is better
            // updated while dW_hid finishes
            SBR_LAUNCH(upd_front(y.layer[0].p_Whid));                     // W_in, b: main-stream gradients only
            { const int rc = side_join(h); if (rc != SBR_OK) return rc; }
            SBR_LAUNCH(upd(y.layer[0].p_Whid, p_end));                    // W_hid, peepholes, initial states, output layer
        } else {
            SBR_HIP(hipStreamWaitEvent(h->stream, h->ev_og, 0));
            size_t pos = 0;
            int l_from = 0;
            if (row_aware) {      // layer 0: the row-aware pass over W_in, then b
                SBR_LAUNCH(upd_front(y.layer[0].p_Whid));
                pos = y.layer[0].p_peep; l_from = 1;
            }
            for (int l = l_from; l < y.L * y.D; ++l) { SBR_LAUNCH(upd(pos, y.layer[l].p_Whid)); pos = y.layer[l].p_peep; }
            SBR_LAUNCH(upd(pos, p_end));
            { const int rc = side_join(h); if (rc != SBR_OK) return rc; }
            for (int l = 0; l < y.L * y.D; ++l) SBR_LAUNCH(upd(y.layer[l].p_Whid, y.layer[l].p_peep));
        }
    } else {
        { const int rc = side_join(h); if (rc != SBR_OK) return rc; }
        SBR_LAUNCH(upd_front(p_end));
    }
    h->og_recorded = false; h->tail_swapped = false; h->tail_updated = false; h->out_early = false;
    h->wout_early = false;
    mark(h, 7);
    if (!h->in_train_step && h->timing) h->ring_used += 1;
    h->grads_clean = true;
    h->fwd_done = false;
    h->train_fwd_open = false;      // the step is complete and its streams are joined: the next batch build may trust main-stream order
    if (h->side_pending || h->tail_join_pending) h->bb_slow = 2;
    return SBR_OK;
}

// The recurrent kernels' bounded spin-waits raise a flag instead of hanging the GPU.  Every call that hands results to the
// host checks it (training: with the cost; inference: with the ids / scores) and CLEARS it, so that one timeout fails
// the call it belongs to and not every later call of the handle.
static int report_fault(sbr_handle* h, int fault) {
    if (!fault) return SBR_OK;
    (void)hipMemsetAsync(h->A(h->lay.a_fault), 0, sizeof(int), h->stream);
    // bit 0: cluster exchange (sbr_rec_cl.hip); bits 1, 2: publish counter / pipe gate of the pipelined kernels (sbr_rec_p.hip);
    // bit 3: a consumer of the overlapped tail (or its monitor) waited for the chain for 1.5 s; bit 4: a unit of the LDS-row
    // scatter-add was handed more ids than it has LDS rows for (launch_scatter_lds_poll sizes them: cannot happen)
    if (fault & 16)
        sbr_set_error("the LDS-row scatter-add of the overlapped tail ran out of rows (flag %d, results of this call invalid); rerun with "
                      "SBR_TAIL_SCATTER_LDS=0", fault);
    else
    sbr_set_error("a bounded wait inside the recurrent kernels gave up (flag %d, results of this call invalid); rerun with %s", fault,
                  (fault & 1) ? "SBR_CLUSTER=0" : (fault & 8) ? "SBR_TAIL_OVERLAP=0" : "SBR_X6_PIPE=0");
    return SBR_EHIP;
}
static int check_fault(sbr_handle* h) {        // synchronises the stream
    int fault = 0;
    SBR_HIP(hipMemcpyAsync(&fault, h->A(h->lay.a_fault), sizeof(int), hipMemcpyDeviceToHost, h->stream));
    SBR_HIP(hipStreamSynchronize(h->stream));
    return report_fault(h, fault);
}

extern "C" int sbr_read_cost(sbr_handle* h, float* cost_host) {
    CHECK_ARG(h && cost_host, "null argument");
    SBR_HIP(hipMemcpyAsync(cost_host, h->cost_ptr(), sizeof(float), hipMemcpyDeviceToHost, h->stream));
    return check_fault(h);
}

extern "C" int sbr_train_step(sbr_handle* h, float* cost_host) {
    CHECK_ARG(h, "null handle");
    int rc;
    if (h->timing) h->ring_cur = h->ring_used % sbr_handle::kRing;
    h->in_train_step = true;
    struct Guard { sbr_handle* h; ~Guard() { h->in_train_step = false; } } guard{h};
    mark(h, 0);
    if ((rc = sbr_zero_grads(h)) != SBR_OK) return rc;
    if ((rc = sbr_forward(h)) != SBR_OK) return rc;
    if ((rc = sbr_loss_backward_output(h)) != SBR_OK) return rc;
    if ((rc = sbr_backward_recurrent(h)) != SBR_OK) return rc;
    // train_function returns the cost of the batch BEFORE the update (rnn_base.py:290)
    if ((rc = sbr_apply_update(h)) != SBR_OK) return rc;
    if (h->timing) h->ring_used += 1;
    if (cost_host) return sbr_read_cost(h, cost_host);
    return SBR_OK;
}

// The cost and the fault word of a lagged step reach the host through ONE one-thread kernel that stores them into pinned host memory
// and then a sequence number (round 6; before: two 4-byte device-to-host copies and an event record on the main stream between two
// steps, ~10 us of the training loop at C2).  The host reads them one step later: the number is there long before.
__global__ void lag_report_kernel(const float* cost, const int* fault, volatile float* host, int slot, unsigned seq) {
    host[slot] = *cost;
    ((volatile int*)host)[2 + slot] = *fault;
    __threadfence_system();
    ((volatile unsigned*)host)[4 + slot] = seq;
}

static int lagged_collect(sbr_handle* h, float* cost, int* have) {
    *have = 0;
    if (h->lag_pending < 0) return SBR_OK;
    const int s = h->lag_pending;
    h->lag_pending = -1;
    volatile unsigned* q = (volatile unsigned*)&h->lag_host[4 + s];
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; *q != h->lag_seq[s]; ++spins) {
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
            SBR_HIP(hipStreamSynchronize(h->stream));      // (a step that long, or a failed one: the stream says which)
            if (*q != h->lag_seq[s]) { sbr_set_error("lagged step: its report never arrived"); return SBR_EHIP; }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *cost = h->lag_host[s];
    *have = 1;
    int fault = 0;
    memcpy(&fault, (const void*)&h->lag_host[2 + s], sizeof(int));
    return report_fault(h, fault);
}

extern "C" int sbr_train_step_lagged(sbr_handle* h, float* prev_cost, int* have_prev) {
    CHECK_ARG(h && prev_cost && have_prev, "null argument");
    int rc = sbr_train_step(h, nullptr);
    if (rc != SBR_OK) return rc;
    const int s = h->lag_slot;
    h->lag_seq[s] = ++h->lag_counter;
    lag_report_kernel<<<1, 1, 0, h->stream>>>(h->cost_ptr(), (const int*)h->A(h->lay.a_fault), h->lag_host, s, h->lag_seq[s]);
    SBR_LAUNCH(hipGetLastError());
    rc = lagged_collect(h, prev_cost, have_prev);          // the step before this one: normally long finished
    h->lag_pending = s;
    h->lag_slot = s ^ 1;
    return rc;
}

extern "C" int sbr_lagged_flush(sbr_handle* h, float* cost, int* have) {
    CHECK_ARG(h && cost && have, "null argument");
    return lagged_collect(h, cost, have);
}

// ---------------------------------------------------------------------------------------
// predict / top-k
// ---------------------------------------------------------------------------------------
static int full_scores(sbr_handle* h, int do_softmax) {
    const Layout& y = h->lay;
    int rc;
    if (!h->fwd_done && (rc = sbr_forward(h)) != SBR_OK) return rc;
    if ((rc = flush_lazy(h, 1)) != SBR_OK) return rc;      // every item is scored: all of W_out^T / b_out must be current
    float* lg = h->A(y.a_logits);
    // scoring always runs the exact-f32 kernel: a row's scores (hence its ranked ids) must not depend on how many rows
    // share the call (the bf16x6 kernel takes over at >= 96 rows and rounds differently)
    // (SBR_FLAG_BF16_PROJECTION: the single-plane bf16 kernel, which serves any number of rows with the same arithmetic)
    sbr_gemm_set_exact_f32(true);
    const bool bf16p = (y.cfg.flags & SBR_FLAG_BF16_PROJECTION) && !simple_gemm(h);
    if (bf16p) sbr_gemm_set_planes(1);
    const hipError_t ge = launch_gemm(h->stream, h_last(h), y.HLt, 1, h->P(y.p_WoutT), 1, y.HLt, lg, y.N, h->n_rows, y.N, y.HLt, nullptr,
                                      nullptr, 0, simple_gemm(h));
    sbr_gemm_set_planes(3);
    SBR_LAUNCH(ge);
    SBR_LAUNCH(launch_softmax_rows(h->stream, lg, h->P(y.p_bout), h->n_rows, y.N, do_softmax));
    return SBR_OK;
}

extern "C" int sbr_predict_scores(sbr_handle* h, int probs, float* out_host) {
    CHECK_ARG(h, "null handle");
    if (!h->have_batch) { sbr_set_error("sbr_predict_scores: no batch set"); return SBR_ESTATE; }
    const Layout& y = h->lay;
    const int rc = full_scores(h, (probs || y.cfg.loss == SBR_LOSS_CCE) ? 1 : 0);
    if (rc != SBR_OK) return rc;
    if (out_host) {
        SBR_HIP(hipMemcpyAsync(out_host, h->A(y.a_logits), (size_t)h->n_rows * y.N * sizeof(float), hipMemcpyDeviceToHost, h->stream));
        return check_fault(h);      // a forward that gave up must not hand out scores
    }
    return SBR_OK;
}

extern "C" int sbr_topk(sbr_handle* h, int k, int exclude_seen, int32_t* ids_host) {
    CHECK_ARG(h && ids_host, "null argument");
    if (!h->have_batch) { sbr_set_error("sbr_topk: no batch set"); return SBR_ESTATE; }
    const Layout& y = h->lay;
    CHECK_ARG(k >= 1 && k <= 64 && k <= y.N, "k=%d outside [1,min(64,N)]", k);
    // softmax is monotone: ranking the biased logits == ranking softmax(logits)*(1-exclude)
    // (rnn_base.py:200-207) whenever at least k items are not excluded.
    const int rc = full_scores(h, 0);
    if (rc != SBR_OK) return rc;
    float* lg = h->A(y.a_logits);
    // exclude_seen 1: viewed items can never be ranked (top_k_recommendations, rnn_base.py:154-155); 2: the compiled test
    // function's scores * (1 - exclude) (:201-202) -- the same ranking for probabilities, NOT for RNNMargin's raw outputs,
    // where a viewed item then scores 0 and outranks every negative one
    if (exclude_seen)
        SBR_LAUNCH(launch_exclude_seen(h->stream, lg, h->bX, h->blen, h->n_rows, y.T, y.F, y.N,
                                       (exclude_seen == 2 && SBR_LOSS_IS_MARGIN(y.cfg.loss)) ? 0.0f : -INFINITY));
    int* ids = (int*)h->A(y.a_topk);
    SBR_LAUNCH(launch_topk(h->stream, lg, h->n_rows, y.N, k, ids));
    SBR_HIP(hipMemcpyAsync(ids_host, ids, (size_t)h->n_rows * k * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    return check_fault(h);          // ... nor rankings (test.py, validation)
}

// ---------------------------------------------------------------------------------------
// debug / timing
// ---------------------------------------------------------------------------------------
extern "C" int sbr_debug_buffer(sbr_handle* h, const char* name, void** dev_ptr, size_t* n_floats) {
    CHECK_ARG(h && name && dev_ptr && n_floats, "null argument");
    const Layout& y = h->lay;
    const std::string nm(name);
    const size_t tb = (size_t)y.T * y.Bp;
    if (nm == "h_last") { *dev_ptr = h_last(h); *n_floats = (size_t)y.Bp * y.HLt; return SBR_OK; }
    if (nm == "logits") { *dev_ptr = h->A(y.a_logits); *n_floats = (size_t)y.Bp * ((y.N + 3) & ~3); return SBR_OK; }
    if (nm == "dh_last") { *dev_ptr = h->A(y.a_dhlast); *n_floats = (size_t)y.Bp * y.HLt; return SBR_OK; }
    if (nm == "batch_X") { *dev_ptr = (void*)h->bX; *n_floats = (size_t)y.Bp * y.T * y.F; return SBR_OK; }
    if (nm == "batch_lengths") { *dev_ptr = (void*)h->blen; *n_floats = y.Bp; return SBR_OK; }
    if (nm == "batch_target") { *dev_ptr = (void*)h->btgt; *n_floats = y.S > 0 ? y.Bg : (size_t)y.Bp * y.NT; return SBR_OK; }
    if (nm == "batch_pop") { *dev_ptr = (void*)h->bpop; *n_floats = y.Bp; return SBR_OK; }
    if (nm == "batch_samples") { *dev_ptr = (void*)h->bsmp; *n_floats = y.S; return SBR_OK; }
    if (nm == "prof") { *dev_ptr = h->A(y.a_prof); *n_floats = (size_t)2 * (y.Bp / 16) * 16 * 8 * 2; return SBR_OK; }
    if (nm == "prof_head") { *dev_ptr = h->A(y.a_prof) + (size_t)2 * (y.Bp / 16) * 16 * 8 * 2; *n_floats = (size_t)(y.Bp / 16) * 16 * 8 * 2; return SBR_OK; }
    if (nm == "tail_trace" && h->tail_trace) { *dev_ptr = h->tail_trace; *n_floats = 2 * 16384; return SBR_OK; }
    if (nm == "tail_chain_clock") { *dev_ptr = (int*)h->A(y.a_prog) + (y.Bp / h->rpt) * 8 + 128; *n_floats = 8; return SBR_OK; }
    if (nm == "rowcost") { *dev_ptr = h->A(y.a_rowcost); *n_floats = y.Bp; return SBR_OK; }
    if (nm == "act" && y.S > 0) { *dev_ptr = h->A(y.a_act); *n_floats = (size_t)y.Bp * y.C; return SBR_OK; }
    for (int l = 0; l < y.L; ++l) {
        const LayerLayout& ly = y.layer[l];
        const std::string sfx = std::to_string(l);
        if (nm == "xt" + sfx) { *dev_ptr = h->A(ly.a_xt); *n_floats = tb * y.G * ly.Hp; return SBR_OK; }
        if (nm == "hs" + sfx) { *dev_ptr = h->A(ly.a_hs); *n_floats = (tb + y.Bp) * ly.Hp; return SBR_OK; }
        if (nm == "dxt" + sfx) { *dev_ptr = h->A(ly.a_dxt); *n_floats = tb * y.G * ly.Hp; return SBR_OK; }
        if (nm == "dhi" + sfx) { *dev_ptr = h->A(ly.a_dhi); *n_floats = tb * (y.cfg.cell == SBR_CELL_GRU ? 1 : y.G) * ly.Hp; return SBR_OK; }
    }
    sbr_set_error("unknown debug buffer '%s'", name);
    return SBR_EINVAL;
}

extern "C" int sbr_copy_to_host(sbr_handle* h, const void* dev_ptr, float* host, size_t n_floats) {
    CHECK_ARG(h && dev_ptr && host, "null argument");
    SBR_HIP(hipMemcpyAsync(host, dev_ptr, n_floats * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    SBR_HIP(hipStreamSynchronize(h->stream));
    return SBR_OK;
}

extern "C" int sbr_synchronize(sbr_handle* h) {
    CHECK_ARG(h, "null handle");
    SBR_HIP(hipStreamSynchronize(h->stream));
    return SBR_OK;
}

extern "C" int sbr_debug_gemm(void* stream, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                              float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, float* ws,
                              size_t ws_floats, int32_t exact_f32) {
    CHECK_ARG(A && B && C, "null operand");
    sbr_gemm_set_exact_f32(exact_f32 == 1);
    sbr_gemm_set_planes(exact_f32 == 2 ? 1 : 3);
    sbr_gemm_set_planes((exact_f32 == 2 || exact_f32 == 5) ? 1 : 3);
    if (exact_f32 == 3 || exact_f32 == 4) sbr_gemm_hint(2, 1.0f, 1.0f);      // the two-plane fp16 split (three MFMAs): what the step's logits GEMM takes
    sbr_gemm_x6_no_wide(exact_f32 == 4 || exact_f32 == 5);
    const hipError_t ge = launch_gemm((hipStream_t)stream, A, (long)sam, (long)sak, B, (long)sbk, (long)sbn, C, (long)ldc, M, N, K, bias, ws,
                                      ws_floats, false);
    sbr_gemm_x6_no_wide(false);
    sbr_gemm_set_planes(3);
    SBR_LAUNCH(ge);
    return SBR_OK;
}

// a foreign kernel that holds CUs for a while (include/sbr_rnn.h: test hook)
__global__ void __launch_bounds__(256) occupy_kernel(unsigned long long ticks, int* sink) {
    extern __shared__ int occ_lds[];
    occ_lds[threadIdx.x] = (int)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (occ_lds[(threadIdx.x + 1) & 255] == -7) *sink = 1;      // (keeps the LDS claim alive)
}
extern "C" int sbr_debug_occupy(sbr_handle* h, int workgroups, int lds_kb, int milliseconds) {
    CHECK_ARG(h && workgroups >= 0 && workgroups <= 4096 && lds_kb >= 0 && lds_kb <= 160 && milliseconds >= 0 && milliseconds <= 10000, "bad argument");
    static hipStream_t occ = nullptr;
    if (!occ) SBR_HIP(hipStreamCreateWithFlags(&occ, hipStreamNonBlocking));
    if (workgroups == 0) { SBR_HIP(hipStreamSynchronize(occ)); return SBR_OK; }
    const size_t lds = std::max<size_t>(1024, (size_t)lds_kb * 1024);
    SBR_DYN_LDS(occupy_kernel, lds);
    occupy_kernel<<<workgroups, 256, lds, occ>>>((unsigned long long)milliseconds * 100000ull, (int*)h->A(h->lay.a_fault) + 1);
    SBR_LAUNCH(hipGetLastError());
    return SBR_OK;
}

// the stand-alone scatter-add of layer 0's embedding gradient over the current batch, timed on its own (include/sbr_rnn.h)
extern "C" int sbr_debug_scatter(sbr_handle* h, int reps, float* us, int64_t* entries, int64_t* rows) {
    CHECK_ARG(h && us && reps >= 1 && reps <= 1000, "bad argument");
    const Layout& y = h->lay;
    if (!h->have_batch || y.E || y.D != 1) { sbr_set_error("sbr_debug_scatter: needs a batch and an index-input, one-direction layer 0"); return SBR_ESTATE; }
    const LayerLayout& ly = y.layer[0];
    const int GHp = y.G * ly.Hp;
    hipStream_t s = h->stream;
    SBR_HIP(hipDeviceSynchronize());                      // nothing beside it
    SBR_LAUNCH(launch_scatter_sort(s, h->bX, h->blen, y.T, y.Bp, y.F, y.cfg.input_size, (int*)h->A(y.a_scnt), (int*)h->A(y.a_soff),
                                   (int*)h->A(y.a_scur), (int*)h->A(y.a_sid), (int*)h->A(y.a_spos), 0, 0, 1, nullptr, &h->scnt_zero_n));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    SBR_HIP(hipEventCreate(&e0)); SBR_HIP(hipEventCreate(&e1));
    float* dW = h->Gd(ly.p_Win);
    const float* dxt = h->A(ly.a_dxt);
    static const int range_on = [] { const char* e = getenv("SBR_SCAT_RANGE"); return e ? atoi(e) : 1; }();
    auto one = [&]() -> int {
        hipError_t se = hipSuccess;
        if (range_on == 1 && y.a_srpart && GHp <= 1024 &&
            launch_scatter_range(s, dW, dxt, (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos), (const int*)h->A(y.a_soff),
                                 y.cfg.input_size, GHp, h->A(y.a_srpart), (int*)h->A(y.a_srid), SBR_SCAT_RANGES, &se)) { SBR_LAUNCH(se); }
        else if (range_on == 2 && y.a_srpart &&
                 launch_scatter_wide(s, dW, dxt, (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos), (const int*)h->A(y.a_soff),
                                     y.cfg.input_size, y.T * y.Bp * y.F, GHp, h->A(y.a_srpart), (int*)h->A(y.a_srid), y.sr_slots, &se)) { SBR_LAUNCH(se); }
        else SBR_LAUNCH(launch_scatter_reduce(s, dW, dxt, (const int*)h->A(y.a_sid), (const int*)h->A(y.a_spos), (const int*)h->A(y.a_soff),
                                              y.cfg.input_size, y.T * y.Bp * y.F, GHp, y.Bp));
        return SBR_OK;
    };
    for (int i = 0; i < 2; ++i) { const int rc = one(); if (rc != SBR_OK) return rc; }
    SBR_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) { const int rc = one(); if (rc != SBR_OK) return rc; }
    SBR_HIP(hipEventRecord(e1, s));
    SBR_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SBR_HIP(hipEventElapsedTime(&ms, e0, e1));
    *us = ms * 1000.0f / (float)reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (entries || rows) {
        std::vector<int> off((size_t)y.cfg.input_size + 1);
        SBR_HIP(hipMemcpy(off.data(), h->A(y.a_soff), off.size() * sizeof(int), hipMemcpyDeviceToHost));
        int64_t nr = 0;
        for (int i = 0; i < y.cfg.input_size; ++i) nr += off[i + 1] > off[i];
        if (entries) *entries = off[y.cfg.input_size];
        if (rows) *rows = nr;
    }
    // the gradient block as a step expects it: zero (the rows the scatter-add wrote, i.e. the whole block)
    SBR_HIP(hipMemsetAsync(dW, 0, (size_t)y.cfg.input_size * GHp * sizeof(float), s));
    SBR_HIP(hipStreamSynchronize(s));
    h->tail_sorted = false;
    return SBR_OK;
}

extern "C" int sbr_query(sbr_handle* h, const char* what, int64_t* value) {
    CHECK_ARG(h && what && value, "null argument");
    const Layout& y = h->lay; const std::string w(what);
    if (w == "fused_gather") {
        RecArgs a = rec_args(h, 0);
        *value = (y.E == 0 && y.F == 1 && h->fuse_gather && sbr_rec_fwd_can_fuse_gather(a, simple_rec(h))) ? 1 : 0;
    } else if (w == "rows_per_workgroup") *value = h->rpt;
    else if (w == "head_fused") {      // would a full batch of a training step take the one-launch head (sbr_head.hip)?  (its column chunks, or 0)
        int cc = 0, cw = 0; size_t lds = 0;
        *value = (h->head_fuse && y.cfg.loss == SBR_LOSS_CCE && !simple_gemm(h) && !(y.cfg.flags & (SBR_FLAG_BF16_PROJECTION | SBR_FLAG_F32_MFMA)) &&
                  y.D == 1 && y.B == y.Bp && sbr_head_plan(y.Bp, y.N, y.HLt, &cc, &cw, &lds) && (size_t)cc * y.Bp * y.HLt <= y.ws_floats) ? cc : 0;
    }
    else if (w == "cluster") { RecArgs a = rec_args(h, (y.L - 1) * y.D); *value = (!simple_rec(h) && sbr_rec_cluster_ok(a)) ? 1 : 0; }
    else if (w == "rec_kernel") {   // family serving the top layer: 0 triage, 1 cluster, 2 x6p (128 units), 3 x6q (32/64), 4 other
        RecArgs a = rec_args(h, (y.L - 1) * y.D);
        *value = simple_rec(h) ? 0 : sbr_rec_cluster_ok(a) ? 1 : sbr_rec_x6p_ok(a) ? 2 : sbr_rec_x6q_ok(a) ? 3 : 4;
    }
    else if (w.rfind("rec_products_", 0) == 0 || w.rfind("rec_rows_", 0) == 0 || w.rfind("rec_workgroups_", 0) == 0) {
        // what the recurrent kernel of the TOP layer issues on the matrix pipe (bench.py: roofline.matrix_pipe / active_cus):
        // products = low-precision MFMA terms per f32 product (bf16x6: 6, fp16x3: 3; 0 = the exact-f32 MFMA kernels, another
        // pipe rate); rows = live batch rows among the 16 columns of an MFMA tile; workgroups = workgroups of the launch
        const bool bwd = w.size() > 4 && w.compare(w.size() - 4, 4, "_bwd") == 0;
        RecArgs a = rec_args(h, (y.L - 1) * y.D);
        const bool cl = !simple_rec(h) && sbr_rec_cluster_ok(a), xp = sbr_rec_x6p_ok(a), xq = sbr_rec_x6q_ok(a);
        const bool x6 = cl || xp || xq || (!a.f32_mfma && (a.Hp == 32 || a.Hp == 64 || a.Hp == 128));
        int products = 0, rows = 16, wgs = y.Bp / 16;
        if (!simple_rec(h) && x6) {
            const char* fe = getenv(bwd ? "SBR_X6_F16_BWD" : "SBR_X6_F16");      // the launchers' own conditions (sbr_rec_p.hip)
            const bool f16 = (xp || cl || xq) && (fe ? atoi(fe) != 0 : true) && (bwd ? (a.clip > 0.0f && a.clip <= 100.0f) : !a.relu);
            products = f16 ? (xp && !cl ? sbr_rec_x6p_f16_terms() : 3) : 6;
            if (cl && sbr_rec_c16_ok(a)) { rows = 16; wgs = (y.Bp / 16) * (a.Hp / 16); }
            else if (cl) { rows = bwd ? sbr_rec_cluster_bwd_rows(a) : SBR_CL_ROWS; wgs = (y.Bp / rows) * (a.Hp == 256 ? 8 : 32); }
            else { rows = a.rpt; wgs = y.Bp / a.rpt; }
        }
        *value = w.rfind("rec_products_", 0) == 0 ? products : w.rfind("rec_rows_", 0) == 0 ? rows : wgs;
    }
    else if (w == "tail_chunks") { int ch = 0; *value = tail_plan(h, &ch); }      // time chunks of the overlapped step tail (0: not taken)
    // ... whose consumers really run on the side streams (SBR_TAIL_OVERLAP=2 keeps them on the main stream: a data-parallel driver
    // must then not order a collective behind a side stream that produces nothing)
    else if (w == "tail_streams") { int ch = 0; *value = (tail_plan(h, &ch) >= 2 && h->tail_overlap == 1) ? 1 : 0; }
    else if (w == "tail_last_steps") { int ch = 0; *value = tail_plan(h, &ch) >= 2 ? h->tail_bounds.lo[1] : 0; }   // time steps of chunk 0 (behind the chain)
    else if (w == "side_stream2") *value = (int64_t)(intptr_t)h->side2;
    else if (w == "tail_chain_cycles" || w == "tail_chain_ticks") {      // last overlapped-tail BPTT launch: shader cycles / 100 MHz ticks
        unsigned long long c[2] = {0, 0};
        const int nwaves = (y.Bp / h->rpt) * 8;
        SBR_HIP(hipStreamSynchronize(h->stream));
        SBR_HIP(hipMemcpy(c, (const int*)h->A(y.a_prog) + nwaves + 128, sizeof(c), hipMemcpyDeviceToHost));
        *value = (int64_t)c[w == "tail_chain_cycles" ? 0 : 1];
    }
    // overlapped tail, phase-by-phase step: the gradient ranges the two consumer streams produce (floats of the gradient section)
    else if (w == "tail_win_lo") *value = (int64_t)y.layer[0].p_Win;
    else if (w == "tail_win_hi") *value = (int64_t)y.layer[0].p_b;
    else if (w == "tail_whid_lo") *value = (int64_t)y.layer[0].p_Whid;
    else if (w == "tail_whid_hi") *value = (int64_t)y.layer[0].p_peep;
    else if (w == "arena_bytes") *value = (int64_t)(y.s_end * sizeof(float));
    else if (w == "sparse_blocks") *value = y.n_sparse;
    else if (w == "adam_table") *value = y.n_at;
    else if (w == "side_stream") *value = (int64_t)(intptr_t)h->side;
    else { sbr_set_error("unknown query '%s'", what); return SBR_EINVAL; }
    return SBR_OK;
}

extern "C" int sbr_enable_timing(sbr_handle* h, int on) {
    CHECK_ARG(h, "null handle");
    if (on)
        for (int r = 0; r < sbr_handle::kRing; ++r)
            for (int i = 0; i < SBR_N_PHASES; ++i) if (!h->ev[r][i]) SBR_HIP(hipEventCreate(&h->ev[r][i]));
    CHECK_ARG(on >= 0 && on <= SBR_N_PHASES, "on = %d: 0 off, 1 every phase, 2 + p only phase p", on);
    h->timing = on != 0; h->ring_used = 0; h->ring_cur = 0;
    h->timing_marks = on == 1 ? 0xffu : on >= 2 ? (3u << (on - 2)) : 0u;      // a phase lies between marks p and p + 1
    return SBR_OK;
}

// Chain-only timing: on = 1 starts collecting an event pair around every launch of a recurrent chain kernel (all layers, both
// directions; up to 256 launches), on = 0 stops and reports.  us[0] / us[1]: device time summed over the forward / backward chain
// launches since the start; n[0] / n[1]: how many launches that was (divide by the steps run in between).  A survey facility
// like sbr_enable_timing(h, 1): the records cost the stream a few microseconds each.
extern "C" int sbr_chain_times(sbr_handle* h, int on, float us[2], int n[2]) {
    CHECK_ARG(h, "null handle");
    if (on) { h->ch_n = 0; h->chain_timing = true; return SBR_OK; }
    CHECK_ARG(us && n, "null argument");
    h->chain_timing = false;
    SBR_HIP(hipDeviceSynchronize());
    us[0] = us[1] = 0.f; n[0] = n[1] = 0;
    for (int r = 0; r < h->ch_n; ++r) {
        float ms = 0.f;
        if (!h->ev_ch[r][0] || !h->ev_ch[r][1] || hipEventElapsedTime(&ms, h->ev_ch[r][0], h->ev_ch[r][1]) != hipSuccess) continue;
        us[h->ch_dir[r] & 1] += ms * 1000.f; n[h->ch_dir[r] & 1] += 1;
    }
    h->ch_n = 0;
    return SBR_OK;
}

// mean over the (up to 64 most recent) train steps recorded since sbr_enable_timing(h, 1)
extern "C" int sbr_phase_times(sbr_handle* h, float us[SBR_N_PHASES]) {
    CHECK_ARG(h && us, "null argument");
    if (!h->timing || h->ring_used == 0) { sbr_set_error("no timed train step recorded"); return SBR_ESTATE; }
    SBR_HIP(hipStreamSynchronize(h->stream));
    const int n = std::min(h->ring_used, (int)sbr_handle::kRing);
    for (int i = 0; i < SBR_N_PHASES; ++i) us[i] = 0.f;
    for (int r = 0; r < n; ++r)
        for (int i = 0; i < SBR_N_PHASES - 1; ++i) {
            float ms = 0.f;
            if (((h->timing_marks >> i) & 3) != 3) continue;          // this phase was not bracketed
            if (hipEventElapsedTime(&ms, h->ev[r][i], h->ev[r][i + 1]) != hipSuccess) ms = 0.f;
            us[i] += ms * 1000.f / n;
        }
    for (int i = 0; i < SBR_N_PHASES - 1; ++i) us[SBR_N_PHASES - 1] += us[i];
    return SBR_OK;
}
