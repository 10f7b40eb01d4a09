// TEMPORARY: placeholders until unet.hip lands (every call fails loudly).
#include "common.h"
#define NOTYET() do { ofd::set_error("UNet engine not built yet"); return OFD_ERR_STATE; } while (0)
extern "C" {
int ofd_unet_create(const ofd_unet_config*, ofd_unet**) { NOTYET(); }
void ofd_unet_destroy(ofd_unet*) {}
int ofd_unet_num_params(const ofd_unet*) { return 0; }
const char* ofd_unet_param_name(const ofd_unet*, int) { return ""; }
size_t ofd_unet_param_numel(const ofd_unet*, int) { return 0; }
int ofd_unet_set_param(ofd_unet*, int, const float*, size_t, void*) { NOTYET(); }
int ofd_unet_prepare(ofd_unet*, void*) { NOTYET(); }
size_t ofd_unet_workspace_bytes(const ofd_unet*, int, int, int) { return 0; }
int ofd_unet_forward(ofd_unet*, const float*, int, const float*, int, const int64_t*, float*, int, int, int, void*, size_t, void*) { NOTYET(); }
int ofd_unet_read_tap(ofd_unet*, const char*, float*, size_t, void*) { NOTYET(); }
int ofd_unet_set_profiling(ofd_unet*, int) { NOTYET(); }
int ofd_unet_prof_count(const ofd_unet*) { return 0; }
const char* ofd_unet_prof_name(const ofd_unet*, int) { return ""; }
int ofd_unet_prof_read(ofd_unet*, int, double*, long long*, double*, double*) { NOTYET(); }
int ofd_unet_prof_reset(ofd_unet*) { NOTYET(); }
int ofd_conv_forward(const ofd_conv_args*, void*) { NOTYET(); }
size_t ofd_conv_gn_partial_count(int, int, int, int) { return 0; }
size_t ofd_conv_weight_elems(int, int, int) { return 0; }
int ofd_conv_weight_prep(const float*, void*, int, int, int, int, float, int, void*) { NOTYET(); }
}
