// oracle/ref_driver_tu.cpp -- TEST INFRASTRUCTURE.
// Compiles the reference driver 3rdparty/sgbm/sgbm.cpp where it lies, for its qauto / qeasy / paste
// functions (sgbm.cpp:30-71,133-137).  `main` becomes an unused static function so that its iio_*
// references (file I/O through iio.c, not compilable here) are dropped by the compiler; the glue
// of that main is restated, line-cited, in ref_harness.cpp.
#define main static __attribute__((unused)) s2p_ref_unused_main
#include "sgbm.cpp"
#undef main
