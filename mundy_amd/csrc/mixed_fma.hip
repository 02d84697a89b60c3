// mixed_fma.hip -- the minimisation classes of mixed.hip (S-E, E-E) compiled with floating-point contraction ON
// (build.py gives this one file -ffp-contract=fast): a*b+c becomes one fused operation wherever the compiler sees one.
// Results then differ from the oracle's in the last bits, which the L-BFGS line searches amplify; the reference's own
// bar for these distances is 1e-4 (UnitTestEllipsoidEllipsoid.cpp:53).  Labelled, never the default: see
// mhip_contact_mixed_set_contraction.
#define MHIP_MIXED_FMA_TU 1
#include "mixed.hip"
