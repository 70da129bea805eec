// Host-only: the generated source of the column-group Hessian module (pcl_kernel_hess_cols.hpp) for the system in bench/config3_inputs.bin,
// straight from pcl_codegen_v4.hpp -- no library rebuild between edits of the code generator.   usage: hc_dump <inputs.bin> <q> [variant] > out.hip
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pcl_codegen_v4.hpp"
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    int32_t h[8];
    if (fread(h, 4, 8, f) != 8) return 4;
    const int d = h[0], m = h[1], n = 2 * d;
    std::vector<double> G0((size_t)n * n), Gj((size_t)m * n * n);
    if (fread(G0.data(), 8, G0.size(), f) != G0.size() || fread(Gj.data(), 8, Gj.size(), f) != Gj.size()) return 5;
    fclose(f);
    const int q = atoi(argv[2]), variant = argc > 3 ? atoi(argv[3]) : 0;
    const pcl_codegen::V4Plan plan = pcl_codegen::make_v4_plan(d, m, G0.data(), 1, Gj.data());
    if (!plan.ok) return 6;
    const std::string src = "#include \"pcl_device_common.hpp\"\n" + pcl_codegen::v4_functions(plan, q, 1, (variant & 7) | 8, true) + "#include \"pcl_kernel_hess_cols.hpp\"\n";
    fwrite(src.data(), 1, src.size(), stdout);
    return 0;
}
