/* main.c -- the hpgmg-fv executable: same two positional arguments as the
 * reference (finite-volume/source/hpgmg-fv.c:152-205) plus runtime switches
 * for what the reference selects with -D flags.  See driver.c. */
#include "hpgmg_fv.h"
int main(int argc, char **argv) { return hpgmg_fv_main(argc, argv); }
