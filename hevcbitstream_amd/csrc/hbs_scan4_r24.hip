/* hbs_scan4_r24.hip -- K12, event-sparse kernel, the 24-row geometry (96 KiB tiles, up to 1024 elements a tile, all four
 * wavefronts on the element batches; hbs::k_scan_extract4_r24): streams of NALs of ~120 to ~450 bytes, which the 48-row
 * geometry's element budget cannot hold and which the data-independent LDS-image kernel ran at 0.25-0.30 of peak.
 * The source is hbs_scan4_impl.h (reference loops: h264_nal.c:38-76, :147-200). */
#define HBS4_ROWS 24
#include "hbs_scan4_impl.h"
