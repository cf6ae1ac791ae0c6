/* hbs_scan4.hip -- K12, event-sparse kernel, the 48-row geometry (192 KiB tiles; hbs::k_scan_extract4) and everything of
 * hbs_scan4_impl.h that exists once: the call's prologue, the count-ahead, the host-side helpers.  The source is hbs_scan4_impl.h. */
#define HBS4_ROWS 48
#include "hbs_scan4_impl.h"
