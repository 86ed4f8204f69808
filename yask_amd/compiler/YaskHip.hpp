// YaskHip.hpp -- the 'cdna4_hip' format-target of the YASK stencil compiler.
//
// A new PrinterBase subclass for intel/yask's compiler (src/compiler/lib/Print.hpp:426-468), selected
// by `yc_solution::set_target("cdna4_hip")` / `yask_compiler.exe -target cdna4_hip`
// (include/yask_compiler_api.hpp:196-198,462-465; registration point src/compiler/lib/Solution.cpp:240-262,
// see Solution.cpp.patch).  Where the reference's YASKCppPrinter (src/compiler/lib/YaskKernel.cpp) emits
// vector-folded AVX/OpenMP code for its CPU runtime, this printer emits, for the MI355X runtime in
// yask_amd/csrc:
//   * the solution metadata (dims, vars, halos, step-slot counts, L1 norms, parts, stages) -- the same
//     facts YASKCppPrinter::print_context bakes into the generated context;
//   * per stencil part a struct with the access groups, the read offsets and `eval(A&)`: the equations
//     rendered against an accessor, which the HIP kernel templates (ykh_device.hpp) instantiate with
//     register-queue / LDS-slab accessors;
//   * when a part is one equation whose off-centre reads form a single linear combination of one
//     (var, step) group, its *linear star form* (`lin[]`, `eval_lin`) for the gather-past /
//     scatter-future kernel (ykh_starlin.hpp).
// Vector folding does not apply: num_vec_elems() == 1, the wavefront lanes span the unit-stride dim.
#pragma once
#include "Print.hpp"

namespace yask {

    class YASKHipPrinter : public PrinterBase {
    protected:
        Stages& _eq_stages;

    public:
        YASKHipPrinter(Solution& stencil, Parts& parts, Stages& eq_stages) :
            PrinterBase(stencil, parts), _eq_stages(eq_stages) { }
        virtual ~YASKHipPrinter() { }

        virtual int num_vec_elems() const { return 1; }
        virtual bool is_folding_efficient() const { return false; }
        virtual void print(ostream& os);
    };

} // namespace yask.
