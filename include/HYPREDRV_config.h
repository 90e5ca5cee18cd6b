/* HYPREDRV_config.h -- build facts of the MI355X implementation (the reference generates
 * this file with CMake: CMakeLists.txt:30-52). */
#ifndef HYPREDRV_CONFIG_HEADER
#define HYPREDRV_CONFIG_HEADER
#define HYPREDRV_VERSION "0.2.0-amd"
#define HYPREDRV_DEVELOP_STRING "hypredrive_amd (MI355X / gfx950)"
#define HYPREDRV_AMD_GFX950 1
#endif
