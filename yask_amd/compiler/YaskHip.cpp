// YaskHip.cpp -- implementation of the 'cdna4_hip' format-target (see YaskHip.hpp).
#include "Print.hpp"
#include "ExprUtils.hpp"
#include "Eqs.hpp"
#include "Solution.hpp"
#include "YaskHip.hpp"

#include <iomanip>
#include <map>
#include <set>
#include <sstream>

namespace yask {

    namespace {

        // Constants are printed exactly like the reference's generated code prints them
        // (CppPrintHelper::format_real, src/compiler/lib/Cpp.cpp:39-52): 15 significant digits.
        string fmt_real(double v) {
            if (double(int(v)) == v)
                return to_string(int(v));
            ostringstream oss;
            oss << setprecision(15) << scientific << v;
            return oss.str();
        }

        struct Group {
            Var* var;
            int dt;          // step offset (0 if no step dim)
            bool has_step;
            vector<int> misc; // const indices of misc dims, in var-dim order
            int dw = 0;       // offset in the outer (4th) domain dim
            bool operator<(const Group& o) const {
                if (var != o.var) return var < o.var;
                if (dt != o.dt) return dt < o.dt;
                if (dw != o.dw) return dw < o.dw;
                return misc < o.misc;
            }
        };
        struct Off { int g, d[3]; };

        // Everything the emitters need to know about the dims.
        struct DimCtx {
            const Dimensions& dims;
            DimCtx(const Dimensions& d) : dims(d) {}
            // The kernels see the three INNER domain dims as x, y, z (index 0..2); with 4 domain dims the outermost one
            // is an outer loop of launches (index -1 here; DIM_OUTER in the generated metadata).
            int n_outer() const { int n = dims._domain_dims.get_num_dims(); return n > 3 ? n - 3 : 0; }
            int domain_idx(const string& n) const {
                int p = dims._domain_dims.lookup_posn(n);
                return p < 0 ? -2 : p - n_outer();
            }
        };

        // Decompose a var point into (group, domain offsets).
        bool point_info(const DimCtx& dc, VarPoint* vp, Group& g, int ofs[3]) {
            g.var = vp->_get_var();
            g.dt = 0;
            g.has_step = false;
            g.misc.clear();
            g.dw = 0;
            ofs[0] = ofs[1] = ofs[2] = 0;
            for (auto& dim : g.var->get_dims()) {
                auto& dn = dim->_get_name();
                auto type = dim->get_type();
                if (type == STEP_INDEX) {
                    auto* p = vp->get_arg_offsets().lookup(dn);
                    if (!p) return false;
                    g.has_step = true;
                    g.dt = *p;
                } else if (type == DOMAIN_INDEX) {
                    auto* p = vp->get_arg_offsets().lookup(dn);
                    if (!p) return false;       // not a simple offset from the index
                    int di = dc.domain_idx(dn);
                    if (di == -1) { g.dw = *p; continue; }      // the outer dim: part of the access group
                    if (di < 0 || di > 2) return false;
                    ofs[di] = *p;
                } else {
                    auto* p = vp->get_arg_consts().lookup(dn);
                    if (!p) return false;
                    g.misc.push_back(*p);
                }
            }
            return true;
        }

        // Linear form of a sub-expression: sum of coef * read(group, offset).
        struct LinForm {
            bool ok = false;
            int g = -1;
            map<vector<int>, double> terms;   // (dx,dy,dz) -> coef
        };

        class HipEmitter : public ExprVisitor {
        public:
            const DimCtx& dc;
            vector<Group> groups;            // access groups of the part being emitted
            vector<Off> reads;               // distinct reads
            vector<unsigned long long> read_wmask;   // per read: bit k = equation k of the part (the one that writes writes[k]) uses it
            int cur_eq = 0;                  // equation being visited = index in writes[] of the group it writes
            bool dup_writes = false;         // two equations of the part write the same group: read_wmask cannot name equations by writes[] index
            vector<int> read_log;            // indices into reads[] in the order they were touched (temps remember their slice of it)
            map<string, vector<int>> temp_reads;     // temp's expr string -> the reads its value depends on
            vector<int> writes;              // groups written
            ostringstream body;              // statements
            map<string, string> memo;        // expr string -> temp name
            int ntemps = 0;
            const Expr* lin_node = 0;        // node to be replaced by `lin_sum` (eval_lin)
            bool failed = false;
            string fail_why;

            HipEmitter(const DimCtx& d) : dc(d) {}

            int group_of(const Group& g) {
                for (size_t i = 0; i < groups.size(); i++)
                    if (!(groups[i] < g) && !(g < groups[i])) return (int)i;
                groups.push_back(g);
                return (int)groups.size() - 1;
            }
            int find_group(const Group& g) const {
                for (size_t i = 0; i < groups.size(); i++)
                    if (!(groups[i] < g) && !(g < groups[i])) return (int)i;
                return -1;
            }
            void note_read(int g, const int* o) {
                const unsigned long long bit = cur_eq < 64 ? 1ull << cur_eq : 0ull;
                for (size_t i = 0; i < reads.size(); i++) {
                    auto& r = reads[i];
                    if (r.g == g && r.d[0] == o[0] && r.d[1] == o[1] && r.d[2] == o[2]) { read_wmask[i] |= bit; read_log.push_back((int)i); return; }
                }
                reads.push_back(Off{g, {o[0], o[1], o[2]}});
                read_wmask.push_back(bit);
                read_log.push_back((int)reads.size() - 1);
            }
            // A sub-expression an EARLIER equation already evaluated into a temp: the current equation uses that temp, hence every read
            // the temp's value depends on (without this a cluster holding only the later equation would see those reads as
            // "not mine" and evaluate the shared temp from zeros).
            bool reuse_temp(const string& key, string& name) {
                auto it = memo.find(key);
                if (it == memo.end()) return false;
                const unsigned long long bit = cur_eq < 64 ? 1ull << cur_eq : 0ull;
                for (int i : temp_reads[key]) { read_wmask[i] |= bit; read_log.push_back(i); }
                name = it->second;
                return true;
            }
            void close_temp(const string& key, size_t mark) { temp_reads[key] = vector<int>(read_log.begin() + mark, read_log.end()); }
            string temp(const string& key, const string& rhs) {
                auto it = memo.find(key);
                if (it != memo.end()) return it->second;
                string n = "e" + to_string(++ntemps);
                body << "        V " << n << " = " << rhs << "; a.pin(" << n << ");\n";
                memo[key] = n;
                return n;
            }
            string fail(const string& why) { failed = true; if (fail_why.empty()) fail_why = why; return "V(0)"; }

            string visit(ConstExpr* ce) override { return "real_t(" + fmt_real(ce->get_num_val()) + ")"; }
            string visit(CodeExpr* ce) override { return fail("hand-written code expression"); }
            string visit(IndexExpr* ie) override {
                auto type = ie->get_type();
                if (type == DOMAIN_INDEX) {
                    if (dc.domain_idx(ie->_get_name()) < 0) return fail("index of the outer (4th) domain dim used as a value");
                    return "a.template idx<" + to_string(dc.domain_idx(ie->_get_name())) + ">()";
                }
                if (type == STEP_INDEX) return "a.step()";
                return fail("misc index used as a value");
            }
            string visit(VarPoint* vp) override {
                if (lin_node == vp) return "lin_sum";
                Group g; int o[3];
                if (!point_info(dc, vp, g, o)) return fail("var index that is not 'dim +/- const'");
                int gi = group_of(g);
                note_read(gi, o);
                return "a.template rd<" + to_string(gi) + ", " + to_string(o[0]) + ", " + to_string(o[1]) + ", " + to_string(o[2]) + ">()";
            }
            string visit(UnaryNumExpr* ue) override {
                if (lin_node == ue) return "lin_sum";
                const string key = ue->make_str();
                string had;
                if (reuse_temp(key, had)) return had;
                const size_t mark = read_log.size();
                string r = ue->_get_rhs()->accept(this);
                string n = temp(key, ue->get_op_str() + "(" + r + ")");
                close_temp(key, mark);
                return n;
            }
            string visit(BinaryNumExpr* be) override {
                if (lin_node == be) return "lin_sum";
                const string key = be->make_str();
                string had;
                if (reuse_temp(key, had)) return had;
                const size_t mark = read_log.size();
                string n = visit_binary(be);
                close_temp(key, mark);
                return n;
            }
            string visit_binary(BinaryNumExpr* be) {
                string l = be->_get_lhs()->accept(this);
                string r = be->_get_rhs()->accept(this);
                if (be->get_op_str() == "%") return fail("modulo operator");
                // Subtraction and division go through the accessor: hipcc has no packed fp32 subtract (a v2f32 fsub is split
                // into two v_sub_f32 where an add or an fma is one v_pk_* instruction) and an IEEE fp32 division is ~11
                // instructions per lane; the kernel families decide how to issue them (default: the plain operators).
                if (be->get_op_str() == "-") return temp(be->make_str(), "a.sub(" + l + ", " + r + ")");
                if (be->get_op_str() == "/") return temp(be->make_str(), "a.div(" + l + ", " + r + ")");
                return temp(be->make_str(), l + " " + be->get_op_str() + " " + r);
            }
            string visit(CommutativeExpr* ce) override {
                if (lin_node == ce) return "lin_sum";
                const string key = ce->make_str();
                string had;
                if (reuse_temp(key, had)) return had;
                const size_t mark = read_log.size();
                string s;
                for (auto& op : ce->get_ops()) {
                    string o = op->accept(this);
                    s += (s.empty() ? "" : " " + ce->get_op_str() + " ") + o;
                }
                string n = temp(key, s);
                close_temp(key, mark);
                return n;
            }
            string visit(FuncExpr* fe) override {
                const string key = fe->make_str();
                string had;
                if (reuse_temp(key, had)) return had;
                const size_t mark = read_log.size();
                string s = "ykh::fn_" + fe->get_op_str() + "(";
                bool first = true;
                for (auto& op : fe->get_ops()) { s += string(first ? "" : ", ") + "V(" + op->accept(this) + ")"; first = false; }
                string n = temp(key, s + ")");
                close_temp(key, mark);
                return n;
            }
            string visit(UnaryNum2BoolExpr*) override { return fail("boolean expression in a value"); }
            string visit(UnaryBoolExpr*) override { return fail("boolean expression in a value"); }
            string visit(BinaryNum2BoolExpr*) override { return fail("boolean expression in a value"); }
            string visit(BinaryBoolExpr*) override { return fail("boolean expression in a value"); }
            string visit(EqualsExpr* ee) override {
                Group g; int o[3];
                VarPoint* lhs = ee->_get_lhs().get();
                if (!point_info(dc, lhs, g, o) || o[0] || o[1] || o[2]) return fail("write that is not at the centre point");
                // the equation's number = the index its written group has (or is about to get) in writes[]; the group itself is only
                // registered AFTER the right-hand side, so that the numbering of the groups stays "in order of first use"
                cur_eq = (int)writes.size();
                {
                    const int known = find_group(g);
                    for (size_t k = 0; known >= 0 && k < writes.size(); k++)
                        if (writes[k] == known) { cur_eq = (int)k; dup_writes = true; }     // a second equation for the same group
                }
                string rhs = ee->_get_rhs()->accept(this);
                int gi = group_of(g);
                bool have = false;
                for (int w : writes) have |= (w == gi);
                if (!have) writes.push_back(gi);
                body << "        a.template wr<" << gi << ">(" << rhs << ");\n";
                return "";
            }
        };


        // Renders sub-domain (IF_DOMAIN) and step (IF_STEP) conditions as integer/boolean C++ expressions over
        // scalar indices: a.sidx<D>() = global index of the point, a.first_idx<D>()/last_idx<D>() = first/last
        // index of the overall domain (yc_node_factory::new_first/last_domain_index), `t` = the step index.
        class CondEmitter : public ExprVisitor {
        public:
            const DimCtx& dc;
            bool step_only;          // IF_STEP: only the step index may appear
            HipEmitter* em;          // non-null: IF_STEP evaluated on the device, may read vars without domain dims
            bool failed = false;
            string fail_why;
            CondEmitter(const DimCtx& d, bool so, HipEmitter* e = 0) : dc(d), step_only(so), em(e) {}
            string fail(const string& why) { failed = true; if (fail_why.empty()) fail_why = why; return "0"; }
            string visit(ConstExpr* ce) override {
                double v = ce->get_num_val();
                if (double((long long)v) == v) return "(long long)" + to_string((long long)v);
                return "(" + fmt_real(v) + ")";
            }
            string visit(CodeExpr*) override { return fail("hand-written code expression in a condition"); }
            string visit(IndexExpr* ie) override {
                auto type = ie->get_type();
                if (type == STEP_INDEX) return (step_only && !em) ? "t" : "a.sstep()";
                if (step_only) return fail("non-step index in a step condition");
                if (dc.domain_idx(ie->_get_name()) == -1) return fail("index of the outer (4th) domain dim in a condition");
                string d = to_string(dc.domain_idx(ie->_get_name()));
                if (type == DOMAIN_INDEX) return "a.template sidx<" + d + ">()";
                if (type == FIRST_INDEX) return "a.template first_idx<" + d + ">()";
                if (type == LAST_INDEX) return "a.template last_idx<" + d + ">()";
                return fail("misc index in a condition");
            }
            string visit(VarPoint* vp) override {
                // Values of vars without step and domain dims (scalars, misc-dim tables at constant indices) may
                // steer a step condition; the kernel then evaluates it (the reference reads them on the host,
                // YaskKernel.cpp "is_in_valid_step").
                if (!em || !step_only) return fail("var value in a condition");
                Group g; int o[3];
                if (!point_info(dc, vp, g, o) || g.has_step) return fail("var with a step index in a condition");
                for (auto& dim : g.var->get_dims())
                    if (dim->get_type() == DOMAIN_INDEX) return fail("var over domain dims in a condition");
                int gi = em->group_of(g);
                em->note_read(gi, o);
                return "(double)a.template rd<" + to_string(gi) + ", 0, 0, 0>()";
            }
            string visit(UnaryNumExpr* ue) override { return "(" + ue->get_op_str() + ue->_get_rhs()->accept(this) + ")"; }
            string visit(UnaryNum2BoolExpr* ue) override { return "(" + ue->get_op_str() + ue->_get_rhs()->accept(this) + ")"; }
            string visit(UnaryBoolExpr* ue) override { return "(" + ue->get_op_str() + ue->_get_rhs()->accept(this) + ")"; }
            string visit(BinaryNumExpr* be) override {
                return "(" + be->_get_lhs()->accept(this) + " " + be->get_op_str() + " " + be->_get_rhs()->accept(this) + ")"; }
            string visit(BinaryNum2BoolExpr* be) override {
                return "(" + be->_get_lhs()->accept(this) + " " + be->get_op_str() + " " + be->_get_rhs()->accept(this) + ")"; }
            string visit(BinaryBoolExpr* be) override {
                return "(" + be->_get_lhs()->accept(this) + " " + be->get_op_str() + " " + be->_get_rhs()->accept(this) + ")"; }
            string visit(CommutativeExpr* ce) override {
                string r;
                for (auto& op : ce->get_ops()) r += (r.empty() ? "" : " " + ce->get_op_str() + " ") + op->accept(this);
                return "(" + r + ")";
            }
            string visit(FuncExpr*) override { return fail("function call in a condition"); }
            string visit(EqualsExpr*) override { return fail("equation in a condition"); }
        };

        // Linear-form analysis (bottom-up, no code generation).
        class LinAnalyzer {
        public:
            const DimCtx& dc;
            HipEmitter& em;     // for group numbering shared with the emitted code
            LinAnalyzer(const DimCtx& d, HipEmitter& e) : dc(d), em(e) {}

            LinForm analyze(Expr* e) {
                LinForm lf;
                if (auto* vp = dynamic_cast<VarPoint*>(e)) {
                    Group g; int o[3];
                    if (!point_info(dc, vp, g, o)) return lf;
                    lf.ok = true;
                    lf.g = em.group_of(g);
                    lf.terms[{o[0], o[1], o[2]}] = 1.0;
                    return lf;
                }
                if (auto* ce = dynamic_cast<CommutativeExpr*>(e)) {
                    if (ce->get_op_str() == "+") {
                        lf.ok = true;
                        for (auto& op : ce->get_ops()) {
                            LinForm s = analyze(op.get());
                            if (!s.ok || (lf.g >= 0 && s.g != lf.g)) { lf.ok = false; return lf; }
                            lf.g = s.g;
                            for (auto& t : s.terms) lf.terms[t.first] += t.second;
                        }
                        return lf;
                    }
                    if (ce->get_op_str() == "*") {
                        double k = 1.0;
                        LinForm inner;
                        int nlin = 0;
                        for (auto& op : ce->get_ops()) {
                            if (op->is_const_val()) k *= op->get_num_val();
                            else { inner = analyze(op.get()); nlin++; }
                        }
                        if (nlin != 1 || !inner.ok) return lf;
                        lf = inner;
                        for (auto& t : lf.terms) t.second *= k;
                        return lf;
                    }
                    return lf;
                }
                if (auto* ue = dynamic_cast<UnaryNumExpr*>(e)) {
                    // note: BinaryNumExpr derives from UnaryNumExpr; test it first
                    if (auto* be = dynamic_cast<BinaryNumExpr*>(e)) {
                        LinForm l = analyze(be->_get_lhs().get());
                        if (be->get_op_str() == "-") {
                            LinForm r = analyze(be->_get_rhs().get());
                            if (!l.ok || !r.ok || l.g != r.g) return lf;
                            lf = l;
                            for (auto& t : r.terms) lf.terms[t.first] -= t.second;
                            return lf;
                        }
                        if (be->get_op_str() == "/" && l.ok && be->_get_rhs()->is_const_val()) {
                            lf = l;
                            for (auto& t : lf.terms) t.second /= be->_get_rhs()->get_num_val();
                            return lf;
                        }
                        return lf;
                    }
                    if (ue->get_op_str() == "-") {
                        lf = analyze(ue->_get_rhs().get());
                        for (auto& t : lf.terms) t.second = -t.second;
                        return lf;
                    }
                }
                return lf;
            }

            static bool off_centre(const LinForm& lf) {
                for (auto& t : lf.terms)
                    if (t.first[0] || t.first[1] || t.first[2]) return true;
                return false;
            }

            // Maximal linear nodes that contain off-centre reads.
            void collect(Expr* e, vector<pair<Expr*, LinForm>>& out) {
                LinForm lf = analyze(e);
                if (lf.ok) {
                    if (off_centre(lf)) out.push_back({e, lf});
                    return;
                }
                if (auto* ce = dynamic_cast<CommutativeExpr*>(e)) { for (auto& op : ce->get_ops()) collect(op.get(), out); return; }
                if (auto* fe = dynamic_cast<FuncExpr*>(e)) { for (auto& op : fe->get_ops()) collect(op.get(), out); return; }
                if (auto* be = dynamic_cast<BinaryNumExpr*>(e)) { collect(be->_get_lhs().get(), out); collect(be->_get_rhs().get(), out); return; }
                if (auto* ue = dynamic_cast<UnaryNumExpr*>(e)) { collect(ue->_get_rhs().get(), out); return; }
            }
        };

        string c_ident(const string& s) {
            string r;
            for (char c : s) r += (isalnum((unsigned char)c) ? c : '_');
            return r;
        }

    } // anon namespace.

    void YASKHipPrinter::print(ostream& os) {
        DimCtx dc(_dims);
        const string sname = _stencil._get_name();
        const int nddims = _dims._domain_dims.get_num_dims();
        if (nddims < 1 || nddims > 4)
            THROW_YASK_EXCEPTION("the 'cdna4_hip' target supports 1 to 4 domain dimensions; solution '" + sname +
                                 "' has " + to_string(nddims));
        const int ebytes = _settings._elem_bytes;
        const string real_t = ebytes == 4 ? "float" : "double";

        os << "// Automatically generated by the YASK stencil compiler, format-target 'cdna4_hip'\n"
              "// (yask_amd/compiler/YaskHip.cpp).  Stencil solution '" << sname << "', " << ebytes << "-byte reals.\n"
              "// DO NOT EDIT: regenerate with `make -C yask_amd/compiler gen`.\n"
              "#pragma once\n#include <hip/hip_runtime.h>\n#include \"ykh_meta.hpp\"\n#include \"ykh_fn.hpp\"\n\n"
              "namespace ykh_gen_" << c_ident(sname) << " {\nusing namespace ykh;\ntypedef " << real_t << " real_t;\n\n";

        // ---- dims: step, domain (outer -> inner), misc.
        vector<string> dnames;
        map<string, int> dim_idx;
        os << "static constexpr DimMeta dims[] = {\n";
        {
            auto add = [&](const string& n, const char* type, int di) {
                os << "    {\"" << n << "\", " << type << ", " << di << "},\n";
                dim_idx[n] = (int)dnames.size();
                dnames.push_back(n);
            };
            add(_dims._step_dim, "DIM_STEP", -1);
            int di = -dc.n_outer();
            for (auto& d : _dims._domain_dims) { add(d._get_name(), di < 0 ? "DIM_OUTER" : "DIM_DOMAIN", di < 0 ? -1 : di); di++; }
            for (auto& d : _dims._misc_dims) add(d._get_name(), "DIM_MISC", -1);
        }
        os << "};\n\n";

        // ---- vars.
        vector<Var*> vlist;
        map<Var*, int> var_idx;
        os << "static constexpr VarMeta vars[] = {\n"
              "    // name ndims dims step_alloc halo_l halo_r misc_first misc_last l1 scratch written\n";
        for (auto gp : _vars) {
            if (!gp->is_needed()) continue;
            var_idx[gp] = (int)vlist.size();
            vlist.push_back(gp);
            int hl[3] = {0, 0, 0}, hr[3] = {0, 0, 0}, ohl = 0, ohr = 0;
            string dl, mf, ml;
            int step_alloc = 0;
            bool got_domain = false;
            for (auto& dim : gp->get_dims()) {
                auto& dn = dim->_get_name();
                dl += (dl.empty() ? "" : ", ") + to_string(dim_idx.at(dn));
                int f = 0, l = 0;
                if (dim->get_type() == DOMAIN_INDEX) {
                    got_domain = true;
                    int di = dc.domain_idx(dn);
                    int l = _settings._halo_size > 0 ? _settings._halo_size : gp->get_halo_size(dn, true);
                    int r = _settings._halo_size > 0 ? _settings._halo_size : gp->get_halo_size(dn, false);
                    if (di < 0) { ohl = l; ohr = r; } else { hl[di] = l; hr[di] = r; }
                } else if (dim->get_type() == STEP_INDEX)
                    step_alloc = gp->get_step_dim_info().step_dim_size;
                else {
                    auto* minp = gp->get_min_indices().lookup(dn);
                    auto* maxp = gp->get_max_indices().lookup(dn);
                    if (minp && maxp) { f = *minp; l = *maxp; }
                }
                mf += (mf.empty() ? "" : ", ") + to_string(f);
                ml += (ml.empty() ? "" : ", ") + to_string(l);
            }
            if (dl.empty()) { dl = "0"; mf = "0"; ml = "0"; }
            os << "    {\"" << gp->_get_name() << "\", " << gp->get_num_dims() << ", {" << dl << "}, " << step_alloc
               << ", {" << hl[0] << ", " << hl[1] << ", " << hl[2] << "}, {" << hr[0] << ", " << hr[1] << ", " << hr[2] << "}, {"
               << mf << "}, {" << ml << "}, " << (got_domain ? gp->get_l1_dist() : 0) << ", "
               << (gp->is_scratch() ? "true" : "false") << ", "
               << (_parts.get_output_vars().count(gp) ? "true" : "false");
            if (ohl || ohr) os << ", " << ohl << ", " << ohr;      // halo in the outer (4th) domain dim
            os << "},\n";
        }
        os << "};\n\n";

        // ---- parts, stage by stage.
        vector<string> part_names;
        vector<string> part_meta;
        map<string, int> part_idx;
        vector<pair<string, vector<int>>> stages;
        vector<bool> stage_is_scratch;
        int stage_no = 0;
        int max_mixed_reads = 0;         // over the parts: distinct reads with two or more non-zero domain offsets (box / plane neighbourhoods)
        for (auto& st : _eq_stages.get_all()) {
            vector<int> members;
            for (auto& part : st->get_parts()) {
                const string pname = part->_get_name();
                CounterVisitor stats;
                part->visit_eqs(&stats);

                HipEmitter em(dc);
                for (auto& eq : part->get_eqs()) eq->accept(&em);
                if (em.failed)
                    THROW_YASK_EXCEPTION("the 'cdna4_hip' target cannot render part '" + pname + "' of solution '" +
                                         sname + "': " + em.fail_why);
                {
                    int mixed = 0;
                    for (auto& r : em.reads) mixed += ((r.d[0] != 0) + (r.d[1] != 0) + (r.d[2] != 0)) >= 2;
                    if (mixed > max_mixed_reads) max_mixed_reads = mixed;
                }

                // sub-domain / step conditions: rendered as predicates over the indices.
                string cond_code = "true", step_cond_code = "true";
                bool has_cond = part->cond.get() != 0, has_step_cond = part->step_cond.get() != 0;
                if (has_cond) {
                    CondEmitter ce(dc, false);
                    cond_code = part->cond->accept(&ce);
                    if (ce.failed)
                        THROW_YASK_EXCEPTION("the 'cdna4_hip' target cannot render the sub-domain condition of part '" + pname +
                                             "' of solution '" + sname + "': " + ce.fail_why);
                }
                bool step_cond_dev = false;          // evaluated by the kernel (reads var values)
                string step_cond_dev_code = "true";
                if (has_step_cond) {
                    CondEmitter ce(dc, true);
                    step_cond_code = part->step_cond->accept(&ce);
                    if (ce.failed && ce.fail_why == "var value in a condition") {
                        CondEmitter cd(dc, true, &em);
                        step_cond_dev_code = part->step_cond->accept(&cd);
                        if (!cd.failed) { ce.failed = false; step_cond_dev = true; step_cond_code = "true"; }
                        else ce.fail_why = cd.fail_why;
                    }
                    if (ce.failed)
                        THROW_YASK_EXCEPTION("the 'cdna4_hip' target cannot render the step condition of part '" + pname +
                                             "' of solution '" + sname + "': " + ce.fail_why);
                }

                // linear star form?
                bool has_lin = false;
                LinForm lin;
                string lin_body;
                if (part->get_eqs().size() == 1 && !has_cond && !has_step_cond) {
                    auto& eq = part->get_eqs().front();
                    HipEmitter em2(dc);
                    em2.groups = em.groups;              // same group numbering
                    LinAnalyzer la(dc, em2);
                    vector<pair<Expr*, LinForm>> nodes;
                    la.collect(eq->_get_rhs().get(), nodes);
                    if (nodes.size() == 1) {
                        lin = nodes[0].second;
                        bool axis = true;
                        for (auto& t : lin.terms)
                            if ((t.first[0] != 0) + (t.first[1] != 0) + (t.first[2] != 0) > 1) axis = false;
                        // the star group must not be read off-centre anywhere else and must use all domain dims
                        if (axis && em2.groups.size() == em.groups.size()) {
                            em2.lin_node = nodes[0].first;
                            eq->accept(&em2);
                            bool ok = !em2.failed;
                            for (auto& r : em2.reads)
                                if (r.d[0] || r.d[1] || r.d[2]) ok = false;
                            if (ok) { has_lin = true; lin_body = em2.body.str(); }
                        }
                    }
                }

                os << "// ////// Stencil part '" << pname << "' (stage '" << st->_get_name() << "'): " << stats.get_num_ops()
                   << " FP operation(s), " << stats.get_num_reads() << " read(s), " << stats.get_num_writes()
                   << " write(s) per point.\n";
                for (auto& eq : part->get_eqs())
                    os << "//   " << eq->make_str() << "\n";
                os << "struct " << pname << " {\n    typedef " << real_t << " real_t;\n";
                os << "    static constexpr int n_groups = " << em.groups.size() << ";\n"
                      "    static constexpr AccessGroup groups[" << em.groups.size() << "] = {\n";
                for (size_t g = 0; g < em.groups.size(); g++) {
                    auto& G = em.groups[g];
                    os << "        {" << var_idx.at(G.var) << ", " << G.dt << ", " << (G.has_step ? "true" : "false") << ", "
                       << G.misc.size() << ", {";
                    for (size_t i = 0; i < G.misc.size(); i++) os << (i ? ", " : "") << G.misc[i];
                    os << "}";
                    if (G.dw) os << ", " << G.dw;        // offset in the outer (4th) domain dim
                    os << "},   // g" << g << ": " << G.var->_get_name();
                    if (G.has_step) os << "(" << _dims._step_dim << (G.dt > 0 ? "+" : "") << (G.dt ? to_string(G.dt) : "") << ")";
                    os << "\n";
                }
                os << "    };\n";
                // which groups are vars over ALL domain dims (they share strides and pads; the marching kernels
                // address them with one common offset, other vars through their own strides)
                os << "    static constexpr bool group_full[" << em.groups.size() << "] = {";
                for (size_t g = 0; g < em.groups.size(); g++) {
                    int nd = 0;
                    for (auto& dim : em.groups[g].var->get_dims())
                        if (dim->get_type() == DOMAIN_INDEX) nd++;
                    os << (g ? ", " : "") << (nd == nddims ? "true" : "false");
                }
                os << "};\n";
                // which of the kernels' domain dims each group's var has (bit 0 = x, the marching dim ... bit 2 = z, the unit-stride dim;
                // 3-D solutions, else 7): an operand without x is loaded once per block, one without z is one value per row
                os << "    static constexpr unsigned char group_dims[" << em.groups.size() << "] = {";
                for (size_t g = 0; g < em.groups.size(); g++) {
                    int mask = 0;
                    for (auto& dim : em.groups[g].var->get_dims())
                        if (dim->get_type() == DOMAIN_INDEX) {
                            const int di = dc.domain_idx(dim->_get_name());
                            if (di >= 0 && di < 3) mask |= 1 << di;
                        }
                    os << (g ? ", " : "") << (nddims == 3 ? mask : 7);
                }
                os << "};\n";
                os << "    static constexpr int n_reads = " << em.reads.size() << ";\n"
                      "    static constexpr ReadOff reads[" << (em.reads.size() ? em.reads.size() : 1) << "] = {";
                for (size_t i = 0; i < em.reads.size(); i++) {
                    auto& r = em.reads[i];
                    os << (i % 6 == 0 ? "\n        " : " ") << "{" << r.g << ", " << r.d[0] << ", " << r.d[1] << ", " << r.d[2] << "}"
                       << (i + 1 < em.reads.size() ? "," : "");
                }
                if (em.reads.empty()) os << "{0, 0, 0, 0}";
                os << "};\n";
                // which equations use each read (bit k = the equation that writes writes[k]): lets the runtime evaluate a part as
                // several clusters of its equations, each a kernel over the reads it needs (csrc/ykh_subpart.hpp)
                // (has_read_wmask = false -- more than 64 equations, or two equations writing one group -- switches the clusters off:
                //  clusters_legal<P, K>() in csrc/ykh_subpart.hpp; the array is then a one-element dummy so that the name exists)
                const bool wm_ok = em.writes.size() > 1 && em.writes.size() <= 64 && !em.dup_writes;
                if (em.writes.size() > 1) {
                    os << "    static constexpr bool has_read_wmask = " << (wm_ok ? "true" : "false") << ";\n";
                    os << "    static constexpr unsigned long long read_wmask[" << (wm_ok && em.reads.size() ? em.reads.size() : 1) << "] = {";
                    for (size_t i = 0; wm_ok && i < em.reads.size(); i++)
                        os << (i % 12 == 0 ? "\n        " : " ") << "0x" << hex << em.read_wmask[i] << dec << "ull" << (i + 1 < em.reads.size() ? "," : "");
                    if (!wm_ok || em.reads.empty()) os << "0ull";
                    os << "};\n";
                }
                os << "    static constexpr int n_writes = " << em.writes.size() << ";\n"
                      "    static constexpr int writes[" << em.writes.size() << "] = {";
                for (size_t i = 0; i < em.writes.size(); i++) os << (i ? ", " : "") << em.writes[i];
                os << "};\n\n";
                os << "    // `A` supplies rd<group,dx,dy,dz>() / wr<group>() (scalar or z-vector valued).\n"
                      "    template <class A>\n    __device__ __forceinline__ static void eval(A& a) {\n"
                      "        typedef typename A::V V;\n" << em.body.str() << "    }\n";
                if (has_lin) {
                    os << "\n    // Linear star form: the off-centre reads are lin_sum = sum_i lin[i].c * g" << lin.g
                       << "(x+dx, y+dy, z+dz).\n"
                          "    static constexpr bool has_lin = true;\n"
                          "    static constexpr int lin_group = " << lin.g << ";\n"
                          "    static constexpr int n_lin = " << lin.terms.size() << ";\n"
                          "    static constexpr LinTerm lin[" << lin.terms.size() << "] = {";
                    size_t i = 0;
                    for (auto& t : lin.terms) {
                        ostringstream c;
                        c << setprecision(17) << scientific << t.second;
                        os << (i % 3 == 0 ? "\n        " : " ") << "{" << t.first[0] << ", " << t.first[1] << ", " << t.first[2] << ", "
                           << c.str() << "}" << (i + 1 < lin.terms.size() ? "," : "");
                        i++;
                    }
                    os << "};\n    template <class A>\n"
                          "    __device__ __forceinline__ static void eval_lin(A& a, typename A::V lin_sum) {\n"
                          "        typedef typename A::V V;\n" << lin_body << "    }\n";
                } else
                    os << "    static constexpr bool has_lin = false;\n";
                os << "\n    // IF_DOMAIN / IF_STEP conditions of this part" << (has_cond || has_step_cond ? "" : " (none)") << ".\n"
                      "    static constexpr bool has_domain_cond = " << (has_cond ? "true" : "false") << ";\n"
                      "    template <class A>\n    __device__ __forceinline__ static bool cond(const A& a) { return " << cond_code << "; }\n"
                      "    static constexpr bool has_step_cond = " << (has_step_cond ? "true" : "false") << ";\n"
                      "    static bool step_cond(long long t) { return " << step_cond_code << "; }\n"
                      "    // step condition that reads var values: evaluated per launch by the point kernel\n"
                      "    static constexpr bool has_step_cond_dev = " << (step_cond_dev ? "true" : "false") << ";\n"
                      "    template <class A>\n    __device__ __forceinline__ static bool step_cond_dev(const A& a) { return " << step_cond_dev_code << "; }\n";
                os << "};\n\n";

                ostringstream pm;
                pm << "    {\"" << pname << "\", " << pname << "::n_groups, " << pname << "::groups, " << pname << "::n_reads, "
                   << pname << "::reads, " << pname << "::n_writes, " << pname << "::writes,\n     " << stats.get_num_ops() << ", "
                   << stats.get_num_reads() << ", " << stats.get_num_writes() << ", " << stage_no << ", "
                   << (has_cond ? "true" : "false") << ", " << (has_step_cond ? "true" : "false") << ", "
                   << (part->is_scratch() ? "true" : "false") << ", &" << pname << "::step_cond, "
                   << (step_cond_dev ? "true" : "false");
                // the value of the step index enters the arithmetic (or a device-side condition): launches of different
                // steps then differ in more than their base pointers, which the runtime must know before it replays them
                if (em.body.str().find("a.step()") != string::npos || lin_body.find("a.step()") != string::npos ||
                    cond_code.find("a.sstep()") != string::npos || step_cond_dev_code.find("a.sstep()") != string::npos)
                    pm << ", true";
                pm << "},\n";
                part_idx[pname] = (int)part_names.size();
                members.push_back((int)part_names.size());
                part_names.push_back(pname);
                part_meta.push_back(pm.str());
            }
            if (!st->is_scratch() || !members.empty()) {
                stages.push_back({st->_get_name(), members});
                stage_is_scratch.push_back(st->is_scratch());
            }
            stage_no++;
        }

        os << "static constexpr PartMeta parts[] = {\n";
        for (auto& s : part_meta) os << s;
        os << "};\n";
        for (auto& s : stages) {
            os << "static constexpr int " << c_ident(s.first) << "_parts[] = {";
            for (size_t i = 0; i < s.second.size(); i++) os << (i ? ", " : "") << s.second[i];
            os << "};\n";
        }
        os << "static constexpr StageMeta stages[] = {\n";
        for (auto& s : stages) os << "    {\"" << s.first << "\", " << s.second.size() << ", " << c_ident(s.first) << "_parts},\n";
        os << "};\n\n";
        os << "static constexpr SolnMeta soln = {\"" << sname << "\", \"" << _stencil.get_description() << "\", \"cdna4_hip\", "
           << ebytes << ", " << dnames.size() << ", dims, " << (_dims._step_dir < 0 ? -1 : 1) << ", " << vlist.size() << ", vars, "
           << part_names.size() << ", parts, " << stages.size() << ", stages};\n\n"
              "// part list for the kernel registry (stencil_<name>.hip)\n#define YKH_FOR_EACH_PART(M)";
        for (auto& p : part_names) os << " M(" << p << ")";
        os << "\n";
        // Fusion groups: a run of scratch stages and the stage they feed, as ONE list of part types in evaluation order, with the
        // level (stage within the group) of every part -- parts of a level are independent, a level reads what earlier levels wrote.
        // The runtime can evaluate such a group tile by tile with the scratch vars in the LDS (csrc/ykh_fused.hpp) instead of one
        // sweep of the grid per part.  M(list type, level array, first stage index, last stage index).
        {
            ostringstream fg;
            int ngroups = 0;
            size_t k = 0;
            while (k < stages.size()) {
                size_t first = k;
                while (k < stages.size() && stage_is_scratch[k]) k++;
                if (k >= stages.size()) break;            // (scratch stages nobody consumes: none in practice)
                if (k > first) {
                    os << "typedef ykh::PartList<";
                    bool firstp = true;
                    ostringstream lv;
                    for (size_t st = first; st <= k; st++)
                        for (int pi : stages[st].second) {
                            os << (firstp ? "" : ", ") << part_names[pi];
                            lv << (firstp ? "" : ", ") << (st - first);
                            firstp = false;
                        }
                    os << "> fuse_group_" << ngroups << ";\n"
                       << "static constexpr int fuse_group_" << ngroups << "_level[] = {" << lv.str() << "};\n";
                    fg << " M(fuse_group_" << ngroups << ", fuse_group_" << ngroups << "_level, " << first << ", " << k << ")";
                    ngroups++;
                }
                k++;
            }
            os << "#define YKH_FOR_EACH_FUSE_GROUP(M)" << fg.str() << "\n";
        }
        os << "\n}  // namespace ykh_gen_" << c_ident(sname) << "\n";
        // A build hint, read by csrc/Makefile (not by C++ code): the largest number of mixed-offset reads in one part.  Above 8 (MAX_MIXED,
        // csrc/ykh_march.hpp) the registry gives the part plane-ring shapes whose inner loops are long chains of fp32 additions; for
        // those solutions the build adds a second translation unit compiled WITHOUT packed fp32 instructions (v_pk_add_f32 occupies the
        // VALU ~10 cycles per wave against 4 for v_add_f32) whose shapes take part in prepare_solution()'s timing as "_np" twins.
        os << "\n// ykh-build-hint: max-mixed-reads " << max_mixed_reads << "\n";
    }

} // namespace yask.
