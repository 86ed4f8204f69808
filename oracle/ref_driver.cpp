// oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, not product code.
//
// Drives the *unmodified* intel/yask reference CPU kernel library
// (oracle/_ref/lib/libyask_kernel.<stencil>.<arch>.so, built by oracle/Makefile from the sources
// under /root/reference) through its PUBLIC API only (include/yask_kernel_api.hpp:
// yk_factory / yk_env / yk_solution / yk_var), to
//   (1) produce golden outputs from inputs defined by a *logical-index* hash (so that the result
//       does not depend on the reference's folded/padded storage layout), and
//   (2) time run_solution() on the host cores (bench.py's cpu_baseline, kind "reference").
//
// Initial data: every var element inside domain+halo gets
//       value = offset + scale * H(var_ordinal, step_slot, x, y, z),   H in [-1,1)
// with H = yask_amd's "logical hash" (same function in oracle/stencil_oracle.c, the HIP runtime's
// init kernel and yask_amd/hashinit.py). Per-var (offset, scale) come from -init name:offset:scale.
//
// Output (-out PREFIX): PREFIX.json manifest + one raw little-endian file per (var, valid step)
// holding the rank-domain box, row-major in the var's own dim order (last dim fastest).
//
// With -lattice STRIDE the files hold only the sample oracle.py lattice() defines -- every point of the 9-wide boundary layers
// plus every STRIDE-th point per domain dim -- read plane by plane, so that grids of tens of GB (BASELINE config 4's global
// 2048 x 2048 x 1024) never exist twice in memory; the manifest's shape is then the lattice's.  Initialisation goes slab by slab
// along the outermost domain dim for the same reason.
//
// Usage: ref_driver.<tag>.exe -g NX [NY NZ] -steps N [-first T0] [-threads T] [-init v:150:50]...
//                             [-out PREFIX] [-lattice STRIDE] [-trials K] [-opts "<yask options>"] [-reverse]

#include "yask_kernel_api.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

using namespace yask;
using std::string;
using std::vector;

// ---- logical-index hash (keep in sync with oracle/stencil_oracle.c: yo_hash_unit) ----
static inline double hash_unit(int64_t vid, int64_t slot, int64_t x, int64_t y, int64_t z) {
    uint32_t u = (uint32_t)x * 0x9E3779B1u ^ (uint32_t)y * 0x85EBCA77u ^ (uint32_t)z * 0xC2B2AE3Du ^
                 (uint32_t)slot * 0x27D4EB2Fu ^ (uint32_t)vid * 0x165667B1u;
    u ^= u >> 15; u *= 0x2C1B3C6Du; u ^= u >> 12; u *= 0x297A2D39u; u ^= u >> 15;
    return (double)(int32_t)u * (1.0 / 2147483648.0);
}
static inline int64_t imod_flr(int64_t a, int64_t b) { int64_t m = a % b; return m < 0 ? m + b : m; }

struct InitSpec { double offset = 0.0, scale = 1.0; };

int main(int argc, char** argv) {
    vector<idx_t> gsz;
    idx_t nsteps = 1, first_t = 0;
    int threads = 0, trials = 1;
    bool reverse = false;     // run_solution(first, first - steps + 1): step indices descend (reverse-time stencils)
    idx_t lattice_stride = 0; // > 0: -out writes the lattice sample only
    string out_prefix, extra_opts;
    std::map<string, InitSpec> specs;
    for (int i = 1; i < argc; i++) {
        string a = argv[i];
        auto need = [&](int n) { if (i + n >= argc) { std::cerr << "missing value for " << a << "\n"; exit(2); } };
        if (a == "-g") { while (i + 1 < argc && isdigit(argv[i + 1][0])) gsz.push_back(atoll(argv[++i])); }
        else if (a == "-steps") { need(1); nsteps = atoll(argv[++i]); }
        else if (a == "-first") { need(1); first_t = atoll(argv[++i]); }
        else if (a == "-threads") { need(1); threads = atoi(argv[++i]); }
        else if (a == "-trials") { need(1); trials = atoi(argv[++i]); }
        else if (a == "-reverse") reverse = true;
        else if (a == "-out") { need(1); out_prefix = argv[++i]; }
        else if (a == "-lattice") { need(1); lattice_stride = atoll(argv[++i]); }
        else if (a == "-opts") { need(1); extra_opts = argv[++i]; }
        else if (a == "-init") {
            need(1);
            string s = argv[++i];
            auto p1 = s.find(':'), p2 = s.find(':', p1 + 1);
            InitSpec sp;
            sp.offset = atof(s.substr(p1 + 1, p2 - p1 - 1).c_str());
            sp.scale = atof(s.substr(p2 + 1).c_str());
            specs[s.substr(0, p1)] = sp;
        }
        else { std::cerr << "unknown arg " << a << "\n"; return 2; }
    }
    if (gsz.empty()) gsz.push_back(32);

    yk_factory kfac;
    yk_env::disable_debug_output();
    auto env = kfac.new_env();
    auto soln = kfac.new_solution(env);
    auto ddims = soln->get_domain_dim_names();
    string sdim = soln->get_step_dim_name();
    while (gsz.size() < ddims.size()) gsz.push_back(gsz.back());
    for (size_t d = 0; d < ddims.size(); d++) soln->set_overall_domain_size(ddims[d], gsz[d]);
    std::ostringstream opts;
    opts << "-no-auto_tune ";
    if (threads > 0) opts << "-max_threads " << threads << " ";
    opts << extra_opts;
    string rem = soln->apply_command_line_options(opts.str());
    if (!rem.empty()) { std::cerr << "unrecognized yask options: " << rem << "\n"; return 2; }
    soln->prepare_solution();

    const int esz = soln->get_element_bytes();
    auto vars = soln->get_vars();

    auto init_all = [&]() {
        int vid = 0;
        for (auto& v : vars) {
            auto dn = v->get_dim_names();
            const int nd = (int)dn.size();
            InitSpec sp = specs.count(v->get_name()) ? specs[v->get_name()] : InitSpec();
            idx_t_vec first(nd), last(nd);
            int step_posn = -1; idx_t nslots = 1;
            int dom_posn[3] = {-1, -1, -1};   // position of x/y/z-like domain dims (solution order)
            for (int i = 0; i < nd; i++) {
                if (dn[i] == sdim) {
                    step_posn = i;
                    first[i] = v->get_first_valid_step_index(); last[i] = v->get_last_valid_step_index();
                    nslots = last[i] - first[i] + 1;
                } else {
                    bool is_dom = false;
                    // hash coordinates (x, y, z) = the LAST three domain dims; an outer 4th domain dim is folded into the
                    // slot word like a misc dim (that is how the HIP runtime stores it, ykh_meta.hpp DIM_OUTER)
                    const int nouter = ddims.size() > 3 ? (int)ddims.size() - 3 : 0;
                    for (size_t d = 0; d < ddims.size(); d++)
                        if (ddims[d] == dn[i]) { is_dom = true; if ((int)d >= nouter) dom_posn[(int)d - nouter] = i; }
                    if (is_dom) { first[i] = v->get_first_rank_halo_index(dn[i]); last[i] = v->get_last_rank_halo_index(dn[i]); }
                    else { first[i] = v->get_first_misc_index(dn[i]); last[i] = v->get_last_misc_index(dn[i]); }
                }
            }
            // slab by slab along the outermost domain dim (and step by step), at most ~32 M elements in the staging buffer
            const idx_t_vec first_all = first, last_all = last;
            const int split = dom_posn[0] >= 0 ? dom_posn[0] : -1;
            size_t per_plane = 1;
            for (int i = 0; i < nd; i++) if (i != split && i != step_posn) per_plane *= (size_t)(last_all[i] - first_all[i] + 1);
            const idx_t chunk = split >= 0 ? std::max<idx_t>(1, (idx_t)((size_t)(32u << 20) / std::max<size_t>(1, per_plane))) : 1;
            for (idx_t tt = (step_posn >= 0 ? first_all[step_posn] : 0); tt <= (step_posn >= 0 ? last_all[step_posn] : 0); tt++)
            for (idx_t s0 = (split >= 0 ? first_all[split] : 0); s0 <= (split >= 0 ? last_all[split] : 0); s0 += chunk) {
            first = first_all; last = last_all;
            if (step_posn >= 0) first[step_posn] = last[step_posn] = tt;
            if (split >= 0) { first[split] = s0; last[split] = std::min(last_all[split], s0 + chunk - 1); }
            size_t n = 1; vector<idx_t> ext(nd);
            for (int i = 0; i < nd; i++) { ext[i] = last[i] - first[i] + 1; n *= (size_t)ext[i]; }
            vector<double> buf(n);
            vector<idx_t> idx(first.begin(), first.end());
            for (size_t k = 0; k < n; k++) {
                int64_t c[3] = {0, 0, 0};
                for (int d = 0; d < 3; d++) if (dom_posn[d] >= 0) c[d] = idx[dom_posn[d]];
                int64_t slot = step_posn >= 0 ? imod_flr(idx[step_posn], nslots) : 0;
                // misc dims fold into the slot word so each misc index gets distinct data.
                int64_t misc = 0;
                for (int i = 0; i < nd; i++) {
                    bool used = (i == step_posn);
                    for (int d = 0; d < 3; d++) used |= (dom_posn[d] == i);
                    if (!used) misc = misc * 131 + idx[i];
                }
                double h = hash_unit(vid, slot + 1024 * misc, c[0], c[1], c[2]);
                double val = sp.offset + sp.scale * h;
                buf[k] = esz == 4 ? (double)(float)val : val;
                for (int i = nd - 1; i >= 0; i--) { if (++idx[i] <= last[i]) break; idx[i] = first[i]; }
            }
            v->set_elements_in_slice(buf.data(), n, first, last);
            }
            vid++;
        }
    };

    double best = 1e30;
    for (int tr = 0; tr < trials; tr++) {
        init_all();
        auto t0 = std::chrono::steady_clock::now();
        if (nsteps > 0) soln->run_solution(first_t, reverse ? first_t - nsteps + 1 : first_t + nsteps - 1);
        auto t1 = std::chrono::steady_clock::now();
        double sec = std::chrono::duration<double>(t1 - t0).count();
        if (sec < best) best = sec;
        double pts = 1.0; for (auto g : gsz) pts *= (double)g;
        fprintf(stderr, "trial %d: %.6f s, %.4f Gpoints/s\n", tr, sec, nsteps > 0 ? pts * nsteps / sec * 1e-9 : 0.0);
    }
    {
        double pts = 1.0; for (auto g : gsz) pts *= (double)g;
        printf("{\"ref_driver\": \"%s\", \"steps\": %ld, \"best_secs\": %.6f, \"gpoints_per_s\": %.6f, \"threads\": %d, \"target\": \"%s\", \"elem_bytes\": %d}\n",
               soln->get_name().c_str(), (long)nsteps, best, nsteps > 0 ? pts * nsteps / best * 1e-9 : 0.0,
               soln->get_num_outer_threads(), soln->get_target().c_str(), esz);
    }

    if (!out_prefix.empty()) {
        std::ofstream man(out_prefix + ".json");
        man << "{\"solution\": \"" << soln->get_name() << "\", \"elem_bytes\": " << esz << ", \"steps\": " << nsteps
            << ", \"first_step\": " << first_t << ", \"vars\": [";
        bool firstv = true;
        for (auto& v : vars) {
            auto dn = v->get_dim_names();
            const int nd = (int)dn.size();
            idx_t_vec first(nd), last(nd);
            int step_posn = -1;
            for (int i = 0; i < nd; i++) {
                if (dn[i] == sdim) step_posn = i;
                else {
                    bool is_dom = false;
                    for (auto& d : ddims) is_dom |= (d == dn[i]);
                    if (is_dom) { first[i] = v->get_first_rank_domain_index(dn[i]); last[i] = v->get_last_rank_domain_index(dn[i]); }
                    else { first[i] = v->get_first_misc_index(dn[i]); last[i] = v->get_last_misc_index(dn[i]); }
                }
            }
            idx_t t_lo = 0, t_hi = 0;
            if (step_posn >= 0) { t_lo = v->get_first_valid_step_index(); t_hi = v->get_last_valid_step_index(); }
            // (-lattice: only step-indexed vars over three domain dims are sampled -- the wavefields; coefficient vars are inputs)
            if (lattice_stride > 0 && !(step_posn >= 0 && nd == 4)) continue;
            for (idx_t t = t_lo; t <= t_hi; t++) {
                if (step_posn >= 0) first[step_posn] = last[step_posn] = t;
                size_t n = 1; for (int i = 0; i < nd; i++) n *= (size_t)(last[i] - first[i] + 1);
                std::ostringstream fn; fn << out_prefix << "." << v->get_name() << ".t" << t << ".bin";
                std::ofstream f(fn.str(), std::ios::binary);
                vector<idx_t> shape_out;
                for (int i = 0; i < nd; i++) if (i != step_posn) shape_out.push_back(last[i] - first[i] + 1);
                // position of the domain dims among the var's dims, outermost first
                vector<int> dpos;
                for (int i = 0; i < nd; i++) { bool is_dom = false; for (auto& d : ddims) is_dom |= (d == dn[i]); if (is_dom) dpos.push_back(i); }
                if (lattice_stride > 0 && step_posn >= 0 && dpos.size() == 3 && nd == 4) {
                    // the lattice sample only, read plane by plane of the outermost domain dim (oracle.py lattice(): edge 9)
                    auto lat = [&](idx_t lo, idx_t hi) {
                        const idx_t nn = hi - lo + 1, edge = 9;
                        vector<char> take((size_t)nn, 0);
                        for (idx_t k = 0; k < std::min(edge, nn); k++) take[k] = 1;
                        for (idx_t k = 0; k < nn; k += lattice_stride) take[k] = 1;
                        for (idx_t k = std::max<idx_t>(0, nn - edge); k < nn; k++) take[k] = 1;
                        vector<idx_t> out;
                        for (idx_t k = 0; k < nn; k++) if (take[k]) out.push_back(lo + k);
                        return out;
                    };
                    const vector<idx_t> lx = lat(first[dpos[0]], last[dpos[0]]), ly = lat(first[dpos[1]], last[dpos[1]]), lz = lat(first[dpos[2]], last[dpos[2]]);
                    const idx_t y0 = first[dpos[1]], z0 = first[dpos[2]], nz = last[dpos[2]] - first[dpos[2]] + 1;
                    const size_t plane = (size_t)(last[dpos[1]] - first[dpos[1]] + 1) * (size_t)nz;
                    vector<float> pf; vector<double> pd;
                    if (esz == 4) pf.resize(plane); else pd.resize(plane);
                    for (idx_t x : lx) {
                        idx_t_vec pfirst = first, plast = last;
                        pfirst[dpos[0]] = plast[dpos[0]] = x;
                        if (esz == 4) v->get_elements_in_slice(pf.data(), plane, pfirst, plast); else v->get_elements_in_slice(pd.data(), plane, pfirst, plast);
                        for (idx_t y : ly)
                            for (idx_t z : lz) {
                                const size_t o = (size_t)(y - y0) * (size_t)nz + (size_t)(z - z0);
                                if (esz == 4) f.write((char*)&pf[o], 4); else f.write((char*)&pd[o], 8);
                            }
                    }
                    shape_out = {(idx_t)lx.size(), (idx_t)ly.size(), (idx_t)lz.size()};
                }
                else if (esz == 4) { vector<float> b(n); v->get_elements_in_slice(b.data(), n, first, last); f.write((char*)b.data(), n * 4); }
                else { vector<double> b(n); v->get_elements_in_slice(b.data(), n, first, last); f.write((char*)b.data(), n * 8); }
                man << (firstv ? "" : ", ") << "{\"name\": \"" << v->get_name() << "\", \"step\": " << t << ", \"has_step\": "
                    << (step_posn >= 0 ? "true" : "false") << ", \"shape\": [";
                bool f1 = true;
                for (idx_t e : shape_out) { man << (f1 ? "" : ", ") << e; f1 = false; }
                man << "], \"file\": \"" << fn.str().substr(fn.str().find_last_of('/') + 1) << "\"}";
                firstv = false;
            }
        }
        man << "]}\n";
    }
    soln->end_solution();
    env->finalize();
    return 0;
}
