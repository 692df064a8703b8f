# The lines INTEGRATION.md section 2 adds to the reference's CMakeLists.txt for this backend, as an includable module:
#
#   include(<this repo>/cmake/SpblasGfx950.cmake)      # after add_library(spblas INTERFACE), next to the ENABLE_ROCSPARSE block
#
# Pattern: /root/reference/CMakeLists.txt:10 (option) and :80-88 (the rocSPARSE block); the reference-backend condition at
# :97-105 gains `NOT ENABLE_GFX950 AND` (SPBLAS_GFX950_ENABLED below says whether it applies).  Unlike that block no
# vendor sparse library is searched for: the backend is libspblas_gfx950.so (hand-written HIP kernels behind the C ABI of
# include/spblas_gfx950.h) plus the HIP runtime for the stream allocator (vendor/gfx950/stream_memory.hpp).
#
#   SPBLAS_GFX950_ROOT   checkout of this repository (default: the directory above this file)
#   SPBLAS_GFX950_LIBDIR directory of libspblas_gfx950.so (default: <root>/spblas-reference_amd/lib, where
#                        `python -c "import __graft_entry__ as g; g.build()"` puts it)
option(ENABLE_GFX950 "Enable the hand-written MI355X (gfx950) backend" OFF)

set(SPBLAS_GFX950_ENABLED OFF)
if (ENABLE_GFX950)
  if (NOT SPBLAS_GFX950_ROOT)
    get_filename_component(SPBLAS_GFX950_ROOT "${CMAKE_CURRENT_LIST_DIR}/.." ABSOLUTE)
  endif()
  if (NOT SPBLAS_GFX950_LIBDIR)
    set(SPBLAS_GFX950_LIBDIR "${SPBLAS_GFX950_ROOT}/spblas-reference_amd/lib")
  endif()
  if (NOT ROCM_PATH)
    if (DEFINED ENV{ROCM_PATH})
      set(ROCM_PATH "$ENV{ROCM_PATH}")
    else()
      set(ROCM_PATH "/opt/rocm")
    endif()
  endif()
  set(SPBLAS_GPU_BACKEND ON)
  find_library(SPBLAS_GFX950_LIB spblas_gfx950 HINTS "${SPBLAS_GFX950_LIBDIR}" REQUIRED)
  find_library(SPBLAS_GFX950_HIP_RUNTIME amdhip64 HINTS "${ROCM_PATH}/lib" REQUIRED)
  target_include_directories(spblas INTERFACE "${SPBLAS_GFX950_ROOT}/include" "${ROCM_PATH}/include")
  target_link_libraries(spblas INTERFACE "${SPBLAS_GFX950_LIB}" "${SPBLAS_GFX950_HIP_RUNTIME}")
  target_compile_definitions(spblas INTERFACE SPBLAS_ENABLE_GFX950 __HIP_PLATFORM_AMD__)
  set(SPBLAS_GFX950_ENABLED ON)
  message(STATUS "spblas: gfx950 backend enabled (${SPBLAS_GFX950_LIB})")
endif()
