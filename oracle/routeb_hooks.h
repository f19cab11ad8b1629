/* storage hooks of the plugin (include/hpgmg_operators.h), declared for the patched copy of the reference's level.c */
#include <stddef.h>
double *hpgmg_vector_alloc(size_t num_doubles);
void    hpgmg_vector_free(double *p);
void    hpgmg_vector_copy(double *dst, const double *src, size_t num_doubles);
