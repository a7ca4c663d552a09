# oracle/ref/gen_config.cmake -- run with `cmake -P`.
#
# Instantiates the reference's OWN header templates (core/config.h.in,
# compat/compat_config.h.in) with cmake's configure_file(), exactly as its
# build does, for the option set of its "final" OpenGL desktop build
# (CMakeUserPresets.json "rel": CLAP_BUILD_FINAL; core/CMakeLists.txt picks
# CONFIG_RENDERER_OPENGL on Linux).  No header is hand-written: the outputs are
# the reference's templates with these variables substituted.
#
#   cmake -DREF=/root/reference -DOUT=<dir> -P gen_config.cmake
set(CONFIG_RENDERER_OPENGL 1)
set(CONFIG_FINAL 1)
set(HAVE_FFS 1)
set(HAVE_FFSL 1)
set(HAVE_CLOCK_GETTIME 1)
configure_file(${REF}/core/config.h.in ${OUT}/config.h)
configure_file(${REF}/compat/compat_config.h.in ${OUT}/compat_config.h)
