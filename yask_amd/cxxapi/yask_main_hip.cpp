// yask_main_hip.cpp -- the compiled performance / validation harness of the cdna4_hip kernel libraries:
//     bin/yask_kernel.<stencil>.cdna4_hip.exe          (reference: bin/yask_kernel.<stencil>.<arch>.exe)
// Counterpart of the reference's src/kernel/yask_main.cpp:251-665, written against the PUBLIC yk_* C++ API (the
// reference's harness cannot be relinked: it reaches into StencilContext for init_vars / run_ref / compare_data,
// yask_main.cpp:572-616 -- those three go through yk_hip_ext.hpp + the C ABI's extension entry points here).
// Same option names (yask_main.cpp:70-149), same trial protocol, same log keys, so the reference's tooling keeps
// working on the output: src/kernel/yask.sh:595-613 greps `best-throughput`, `mid-throughput`, `TEST PASSED|FAILED`,
// `YASK DONE`; utils/lib/YaskUtils.pm:36-110 parses the `key: value` lines.
//
// Multi-rank: start one process per GPU with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT set (torchrun
// --no-python, mpirun, srun, or yask_amd/bin/yask.sh -ranks N); yk_factory::new_env() does the rest
// (yk_env_init_from_launcher: TCP rendezvous of the ncclUniqueId, RCCL halo transport).
//
// -validate re-runs the trial's steps with the generic one-point-per-thread kernel (-force_scalar) on a second
// solution -- the role of the reference's scalar run_ref() -- and compares with the reference's rule
// (compare_data, epsilon 1e-3, src/kernel/lib/realv.hpp:974-994).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "yask_kernel_api.hpp"
#include "yk_hip_ext.hpp"

using namespace yask;
using std::string;

static const char* DIV = "\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80"
                         "\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80"
                         "\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\xE2\x94\x80\n";

// engineering suffixes like make_num_str (src/common/common_utils.cpp:104-150)
static string num_str(double x) {
    char b[64];
    const double a = std::fabs(x);
    struct { double lim; const char* suf; } up[] = {{1e18, "E"}, {1e15, "P"}, {1e12, "T"}, {1e9, "G"}, {1e6, "M"}, {1e3, "K"}},
                                           dn[] = {{1e-3, "m"}, {1e-6, "u"}, {1e-9, "n"}};
    if (x == 0) return "0";
    for (auto& u : up) if (a >= u.lim) { snprintf(b, sizeof(b), "%g%s", x / u.lim, u.suf); return b; }
    if (a >= 1) { snprintf(b, sizeof(b), "%g", x); return b; }
    for (auto& d : dn) if (a >= d.lim) { snprintf(b, sizeof(b), "%g%s", x / d.lim, d.suf); return b; }
    snprintf(b, sizeof(b), "%g", x);
    return b;
}

struct Opts {
    int num_trials = 3, trial_steps = 10, sleep_secs = 0;
    double trial_time = 0.0, init_seed = 0.1;
    bool warmup = true, pre_auto_tune = false, validate = false, help = false;
    std::vector<string> rest;
};

struct Trial { idx_t nsteps; double secs, pts_ps, reads_ps, writes_ps, flops; };

// The role of init_vars() / set_all_elements_in_seq(-init_seed) (setup.cpp:1023-1040), but independent of the storage
// layout: var k = 1 + k/4 + seed * hash(logical index) -- the same data in both solutions when validating.
static void init_vars(yk_solution_ptr s, double seed) {
    int k = 0;
    for (auto& v : s->get_vars()) {
        if (yk_var_set_elements_hash(yk_hip_handle(v), 1.0 + 0.25 * k, seed, k) != 0) throw yask_exception(yk_last_error());
        k++;
    }
}

int main(int argc, char** argv) {
    Opts o;
    try {
        for (int i = 1; i < argc; i++) {
            string a = argv[i];
            auto val = [&]() -> string {
                if (i + 1 >= argc) throw yask_exception("YASK error: option '" + a + "' requires a value");
                return argv[++i];
            };
            if (a == "-help" || a == "-h" || a == "--help") o.help = true;
            else if (a == "-num_trials" || a == "-t") o.num_trials = std::max(1, atoi(val().c_str()));
            else if (a == "-trial_steps" || a == "-dt") o.trial_steps = std::max(1, atoi(val().c_str()));
            else if (a == "-trial_time") o.trial_time = atof(val().c_str());
            else if (a == "-init_seed") o.init_seed = atof(val().c_str());
            else if (a == "-sleep") o.sleep_secs = atoi(val().c_str());
            else if (a == "-validate" || a == "-v") o.validate = true;
            else if (a == "-warmup") o.warmup = true;
            else if (a == "-no-warmup") o.warmup = false;
            else if (a == "-pre_auto_tune") o.pre_auto_tune = true;
            else if (a == "-no-pre_auto_tune") o.pre_auto_tune = false;
            else o.rest.push_back(a);
        }
        yk_factory kfac;
        auto env = kfac.new_env();
        const int rank = env->get_rank_index(), world = env->get_num_ranks();
        // output from the last rank, like the reference (-msg_rank default)
        std::ostringstream devnull;
        std::ostream& out = (rank == world - 1) ? std::cout : static_cast<std::ostream&>(devnull);
        auto soln = kfac.new_solution(env);
        if (o.help) {
            out << "Usage: " << argv[0] << " [options]\n"
                   " -num_trials|-t <n>  -trial_steps|-dt <n>  -trial_time <secs>  -[no-]warmup  -[no-]pre_auto_tune\n"
                   " -init_seed <x>  -validate|-v  -sleep <secs>\n" << soln->get_command_line_help();
            return 0;
        }
        // (the harness asks for the var-placement search -- a library default since round 3 is "first allocation"; several ranks may
        //  share one device in tests, and the search briefly holds two sets of arrays: one rank per process only)
        if (world == 1) soln->apply_command_line_options("-hip_placement_trials 6");
        string rem = soln->apply_command_line_options(o.rest);
        if (!rem.empty())
            throw yask_exception("YASK error: extraneous parameter(s): '" + rem + "'; run with '-help' option for usage");
        out << DIV << "YASK \xE2\x80\x93 Yet Another Stencil Kit, kernel library " << kfac.get_version_string() << "\n"
            << "Stencil name: " << soln->get_name() << "\nTarget: " << soln->get_target() << "\nElement size: " << soln->get_element_bytes()
            << " bytes\nNum ranks: " << world << "\n";
        soln->prepare_solution();
        yk_soln_h sh = yk_hip_handle(soln);
        out << "Kernel variant(s):";
        for (int p = 0; yk_solution_get_num_kernel_variants(sh, p) > 0; p++) out << " " << yk_solution_get_kernel_variant(sh, p);
        yk_clear_error();
        {   // var placement (-hip_placement_trials): what each set of allocations drawn by prepare_solution() measured
            float pms[32];
            int kept = 0;
            const int np = yk_solution_get_placement_trials(sh, &kept, pms, 32);
            if (np > 0) {
                out << "\nVar placement: " << np << " sets of allocations timed (ms per step):";
                for (int i = 0; i < np && i < 32; i++) out << " " << pms[i];
                out << "; kept set " << kept;
            }
        }
        // the `key: value` lines utils/lib/YaskUtils.pm:36-140 collects (utils/bin/yask_log_to_csv.pl)
        {
            auto dims = soln->get_domain_dim_names();
            auto sizes = [&](const char* key, idx_t (yk_solution::*fn)(const string&) const) {
                out << "\n " << key << ": ";
                for (size_t i = 0; i < dims.size(); i++) out << (i ? " * " : "") << dims[i] << "=" << ((*soln).*fn)(dims[i]);
            };
            idx_t lpts = 1, gpts = 1, bytes = 0;
            for (auto& d : dims) { lpts *= soln->get_rank_domain_size(d); gpts *= soln->get_overall_domain_size(d); }
            for (auto& v : soln->get_vars()) bytes += v->get_num_storage_bytes();
            out << "\n YASK version: " << kfac.get_version_string() << "\n num MPI ranks: " << world
                << "\n num OpenMP threads: 1\n num outer threads: 1\n num inner threads: 1";
            sizes("num-ranks", &yk_solution::get_num_ranks);
            sizes("global-domain size", &yk_solution::get_overall_domain_size);
            sizes("local-domain size", &yk_solution::get_rank_domain_size);
            out << "\n domain size in this rank: " << num_str((double)lpts) << "\n overall problem size: " << num_str((double)gpts)
                << "\n total allocation in this rank: " << num_str((double)bytes) << "B\n total overall allocation: "
                << num_str((double)env->sum_over_ranks(bytes)) << "B\n inner-layout dim: " << dims.back() << "\n inner-loop dim: " << dims.back()
                << "\n num temporal block steps: " << soln->get_block_size(soln->get_step_dim_name()) << "\n";
        }
        init_vars(soln, o.init_seed);
        if (o.pre_auto_tune) {
            out << DIV << "Running the auto-tuner over the compiled tile shapes...\n";
            soln->run_auto_tuner_now(false);
        }
        idx_t steps = o.trial_steps, t0 = 0;
        if (o.warmup) {
            out << DIV << "Running warmup step(s)...\n";
            auto w0 = std::chrono::steady_clock::now();
            soln->run_solution(t0, t0);
            t0 += 1;
            if (o.trial_time > 0) {      // calibrate the number of steps from the warm-up rate
                soln->clear_stats();
                soln->run_solution(t0, t0 + 4);
                t0 += 5;
                double rate = 5.0 / std::max(soln->get_stats()->get_elapsed_secs(), 1e-9);
                steps = std::max<idx_t>(1, (idx_t)(rate * o.trial_time));
                steps = env->sum_over_ranks(steps) / world;       // every rank runs the same number of steps
            }
            out << "  Done in " << num_str(std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count()) << " secs.\n";
        }
        soln->clear_stats();
        const idx_t first_t = t0, last_t = t0 + steps - 1;
        out << DIV << "Running " << o.num_trials << " performance trial(s) of " << steps << " step(s) each...\n";
        std::vector<Trial> trials;
        yk_stats_t last_raw{};
        for (int tr = 0; tr < o.num_trials; tr++) {
            out << DIV << "Trial number:  " << tr + 1 << "\n";
            if (o.validate) init_vars(soln, o.init_seed);
            if (o.sleep_secs > 0) std::this_thread::sleep_for(std::chrono::seconds(o.sleep_secs));
            env->global_barrier();
            soln->clear_stats();
            soln->run_solution(first_t, last_t);
            env->global_barrier();
            if (yk_solution_get_stats(sh, &last_raw) != 0) throw yask_exception(yk_last_error());
            const double secs = last_raw.elapsed_secs;
            Trial t{last_raw.num_steps_done, secs, (double)last_raw.num_elements * last_raw.num_steps_done / secs,
                    last_raw.num_reads_done / secs, last_raw.num_writes_done / secs, last_raw.est_fp_ops_done / secs};
            trials.push_back(t);
            out << " num-steps-done:           " << t.nsteps << "\n elapsed-time (sec):       " << num_str(secs)
                << "\n throughput (num-points/sec): " << num_str(t.pts_ps) << "\n";
            if (world > 1) {
                // time breakdown as the reference prints it (soln_apis.cpp:500-540), from HIP events
                const double comm = last_raw.halo_pack_secs + last_raw.halo_xfer_secs + last_raw.halo_unpack_secs;
                out << " halo-exchange time (sec):   " << num_str(last_raw.halo_secs) << "\n  pack: " << num_str(last_raw.halo_pack_secs)
                    << "  transport: " << num_str(last_raw.halo_xfer_secs) << "  unpack: " << num_str(last_raw.halo_unpack_secs)
                    << "  wait (not hidden): " << num_str(last_raw.halo_wait_secs) << "\n exterior time (sec): " << num_str(last_raw.exterior_secs)
                    << "\n interior time (sec): " << num_str(last_raw.interior_secs) << "\n halo bytes sent per step: "
                    << num_str((double)last_raw.halo_bytes_sent / std::max<idx_t>(1, t.nsteps)) << "\n comm hidden fraction: "
                    << (comm > 0 ? std::max(0.0, 1.0 - last_raw.halo_wait_secs / comm) : 0.0) << "\n";
            }
        }
        std::sort(trials.begin(), trials.end(), [](const Trial& a, const Trial& b) { return a.secs < b.secs; });
        const Trial &best = trials.front(), &mid = trials[trials.size() / 2];
        double sum = 0, sq = 0, mn = 1e300, mx = 0;
        for (auto& t : trials) { sum += t.pts_ps; sq += t.pts_ps * t.pts_ps; mn = std::min(mn, t.pts_ps); mx = std::max(mx, t.pts_ps); }
        const size_t n = trials.size();
        const double sd = n > 2 ? std::sqrt(std::max(0.0, (sq - sum * sum / n) / (n - 1))) : 0.0;
        out << DIV << "Throughput stats across trials:\n"
            << " num-trials:                          " << n << "\n"
            << " min-throughput (num-points/sec):     " << num_str(mn) << "\n"
            << " max-throughput (num-points/sec):     " << num_str(mx) << "\n"
            << " ave-throughput (num-points/sec):     " << num_str(sum / n) << "\n"
            << " std-dev-throughput (num-points/sec): " << num_str(sd) << "\n";
        auto report = [&](const char* tag, const Trial& r, const char* title) {
            const char* pad = string(tag) == "mid" ? " " : "";
            out << DIV << "Performance stats of " << title << ":\n"
                << " " << tag << "-num-steps-done:              " << pad << r.nsteps << "\n"
                << " " << tag << "-elapsed-time (sec):          " << pad << num_str(r.secs) << "\n"
                << " " << tag << "-throughput (num-reads/sec):  " << pad << num_str(r.reads_ps) << "\n"
                << " " << tag << "-throughput (num-writes/sec): " << pad << num_str(r.writes_ps) << "\n"
                << " " << tag << "-throughput (est-FLOPS):      " << pad << num_str(r.flops) << "\n"
                << " " << tag << "-throughput (num-points/sec): " << pad << num_str(r.pts_ps) << "\n";
        };
        report("best", best, "best trial");
        report("mid", mid, "50th-percentile trial");
        bool ok = true;
        if (o.validate) {
            out << "\n" << DIV << "Setup for validation...\n";
            auto ref = kfac.new_solution(env, soln);
            ref->apply_command_line_options(o.rest);
            ref->apply_command_line_options(string("-force_scalar") + (world > 1 ? " -no-overlap_comms -exchange_halos" : ""));
            ref->prepare_solution();
            init_vars(ref, o.init_seed);
            out << "\n" << DIV << "Running " << steps << " step(s) for validation...\n" << std::flush;
            ref->run_solution(first_t, last_t);      // the same steps as the last (re-initialised) trial
            out << "  Done in " << num_str(ref->get_stats()->get_elapsed_secs()) << " secs.\n\nChecking results...\n";
            const idx_t errs = yk_solution_compare_data(sh, yk_hip_handle(ref), 1e-3);
            for (int r = 0; r < world; r++) {
                env->global_barrier();
                if (r == rank) {
                    if (errs == 0) std::cerr << "TEST PASSED on rank " << rank << ".\n";
                    else { std::cerr << "TEST FAILED on rank " << rank << ": " << errs << " mismatch(es).\n"; ok = false; }
                    std::cerr << std::flush;
                }
            }
            ok = env->sum_over_ranks(ok ? 0 : 1) == 0;
            ref->end_solution();
        } else
            out << "\nResults NOT VERIFIED.\n";
        soln->end_solution();
        out << "Stencil '" << soln->get_description() << "'.\n";
        if (!ok) return 1;
        out << "YASK DONE.\n" << DIV << std::flush;
        env->global_barrier();
        return 0;
    } catch (yask_exception& e) {
        std::cerr << "YASK Kernel: " << e.get_message() << ".\n";
        return 1;
    }
}
