/* A plain-C consumer of include/csrk.h: what a compiled host (the reference's own C layer is csr/kernels/mkl/mkl_ops.c)
 * would do with the drop-in library -- plain pointers and sizes, no Python, no torch.  Known-answer inputs from the
 * reference's tests: tests/test_transpose.py:11-27 (rows [0,0,1,3], cols [1,2,0,1], vals 0..3 on a 4 x 3 matrix ->
 * transpose rowptrs [0,1,3,4]) and a product with x = (1, 2, 3).  Exit code 0 = every result as expected. */
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "csrk.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != CSRK_OK) {                                                         \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, csrk_last_error());   \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

int main(void)
{
    /* CSR of the COO above (rows sorted): row 0 = {(1, 0.0), (2, 1.0)}, row 1 = {(0, 2.0)}, row 2 = {}, row 3 = {(1, 3.0)} */
    const int32_t rowptrs[5] = {0, 2, 3, 3, 4}, colinds[4] = {1, 2, 0, 1};
    const double values[4] = {0.0, 1.0, 2.0, 3.0};
    csrk_handle_t h = 0, t = 0;
    CHECK(csrk_create(4, 3, 4, rowptrs, 0, colinds, values, CSRK_VAL_F64, &h));

    const double x[3] = {1.0, 2.0, 3.0};
    double y[4] = {-1.0, -1.0, -1.0, -1.0};
    CHECK(csrk_spmv(h, x, y));
    const double y_ref[4] = {0.0 * 2.0 + 1.0 * 3.0, 2.0 * 1.0, 0.0, 3.0 * 2.0};
    if (memcmp(y, y_ref, sizeof y) != 0) {
        fprintf(stderr, "mult_vec: got %g %g %g %g\n", y[0], y[1], y[2], y[3]);
        return 1;
    }

    int32_t nnzs[4] = {0, 0, 0, 0};
    CHECK(csrk_row_nnzs(h, nnzs));
    int64_t s = -1, e = -1;
    CHECK(csrk_row_extent(h, 3, &s, &e));
    if (nnzs[0] != 2 || nnzs[1] != 1 || nnzs[2] != 0 || nnzs[3] != 1 || s != 3 || e != 4) return 1;

    CHECK(csrk_transpose(h, 1, &t));
    int32_t nr = 0, nc = 0, p64 = -1, vt = -1;
    int64_t nnz = 0;
    CHECK(csrk_info(t, &nr, &nc, &nnz, &p64, &vt));
    if (nr != 3 || nc != 4 || nnz != 4 || p64 != 0 || vt != CSRK_VAL_F64) return 1;
    int32_t trp[4], tci[4];
    double tvs[4];
    CHECK(csrk_export(t, trp, tci, tvs));
    const int32_t trp_ref[4] = {0, 1, 3, 4}, tci_ref[4] = {1, 0, 3, 0};      /* source rows ascend inside a column */
    const double tvs_ref[4] = {2.0, 0.0, 3.0, 1.0};
    if (memcmp(trp, trp_ref, sizeof trp) || memcmp(tci, tci_ref, sizeof tci) || memcmp(tvs, tvs_ref, sizeof tvs)) {
        fprintf(stderr, "transpose: rowptrs %d %d %d %d\n", trp[0], trp[1], trp[2], trp[3]);
        return 1;
    }
    CHECK(csrk_free(t));
    CHECK(csrk_free(h));
    if (csrk_free(h) == CSRK_OK) return 1;      /* a released handle is refused, not dereferenced */
    printf("c consumer ok (libcsrk %d)\n", csrk_version());
    return 0;
}
