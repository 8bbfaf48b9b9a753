// one instantiation of the second-generation SDF kernel for the development harness:
// -DVTAG=name -DVM=mode -DVS=save -DVN=slots -DVD=depth (+ any VDN_SDF2_* tuning macro)
#include "k_sdf_fwd2.h"
#define CAT_(a) sdf2_launch_##a
#define CAT(a) CAT_(a)
extern "C" int CAT(VTAG)(const VdnSdfArgs* args, hipStream_t stream) { return vdn::sdf2::launch<VM, (VS != 0), VN, VD, VARIANT_ID>(args, stream); }
