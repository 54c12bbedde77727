/*
 * TEST INFRASTRUCTURE ONLY (oracle/): exposes two `static` routines of the reference's
 * src/factorization-refinement.c to the tests, the same way the reference's own unit tests reach
 * statics (test/aug_suffix_tree_test.c:10-13 includes the .c under test).  The reference file is
 * compiled from where it lies (-I$(REF)/src); nothing is copied.
 */
#include "factorization-refinement.c"

void ref_static_lcf(const char* s1, size_t l1, const char* s2, size_t l2,
                    size_t* o1, size_t* o2, size_t* len) {
  find_longest_common_factor_dp(s1, l1, s2, l2, o1, o2, len);   /* :255 */
}

int ref_static_longest_affix(char* est, size_t estl, char* gen, size_t genl,
                             size_t* ecut, size_t* gcut) {
  return find_longest_affix(est, estl, gen, genl, ecut, gcut) ? 1 : 0;   /* :1136 */
}
