/* Prototypes of integration/homer_gpu_install.c (compiled inside the HomerHEVC tree, next to hmr_private.h; not part of libhomer_gpu.so). */
#ifndef HOMER_GPU_INSTALL_H
#define HOMER_GPU_INSTALL_H
#ifdef __cplusplus
extern "C" {
#endif
/* route the members of ((hvenc_enc_t *)handle)->funcs to libhomer_gpu.so; want(member_name) != 0 selects a member (NULL: all 19).  Returns how many were routed. */
int hmr_gpu_install(void *homer_handle, int (*want)(const char *member));
/* put back the table HOMER_enc_init had filled */
void hmr_gpu_uninstall(void *homer_handle);
#ifdef __cplusplus
}
#endif
#endif
