# ORACLE build recipe for the REAL reference RANSAC (test infrastructure).
# Compiles degensac + matutls from the reference's own sources where they lie under $(REF) -
# the file lists of degensac/CMakeLists.txt:7-21 and degensac/matutls/CMakeLists.txt:11-27 - into
# oracle/_ref/libdegensac_ref.so.  Nothing is copied into the repo; outputs go to _ref/ only
# (git-ignored, shipped to the GPU box with the snapshot).
# LAPACK (dsyev_/dgesvd_, degensac/lapwrap.c) = the MKL runtime present in the image.
REF ?= /root/reference
CC ?= gcc
OUT = _ref
DEG = DegUtils.c exp_ranF.c exp_ranH.c Ftools.c hash.c Htools.c ranF.c ranH2el.c ranH.c rtools.c utools.c lapwrap.c
MAT = atou1.c cmprt.c hevmax.c mcopy.c qrbdv.c solvru.c trncm.c atovm.c csolv.c hmgen.c minv.c qrecvc.c solvtd.c trnm.c \
      chouse.c cvmul.c house.c mmul.c qreval.c sv2u1v.c unitary.c chousv.c eigen.c housev.c ortho.c qrevec.c sv2uv.c \
      utrncm.c cmattr.c eigval.c ldumat.c otrma.c qrlsq.c sv2val.c utrnhm.c cmcpy.c evmax.c ldvmat.c otrsm.c rmmult.c \
      vmul.c cminv.c hconj.c lsqsv.c psinv.c ruinv.c svdu1v.c cmmul.c heigval.c matprt.c qrbdi.c smgen.c svduv.c \
      cmmult.c heigvec.c mattr.c qrbdu1.c solvps.c svdval.c matconsts.c
MKLDIR ?= /opt/conda/lib
CFLAGS = -O2 -fcommon -fPIC -DA64 -w -I$(REF)/degensac
DEG_OBJS = $(addprefix $(OUT)/obj/deg_,$(DEG:.c=.o))
MAT_OBJS = $(addprefix $(OUT)/obj/mat_,$(MAT:.c=.o))

all: $(OUT)/libdegensac_ref.so

$(OUT)/obj/deg_%.o: $(REF)/degensac/%.c
	@mkdir -p $(OUT)/obj
	$(CC) $(CFLAGS) -c $< -o $@

$(OUT)/obj/mat_%.o: $(REF)/degensac/matutls/%.c
	@mkdir -p $(OUT)/obj
	$(CC) $(CFLAGS) -c $< -o $@

# Two static archives exactly as the reference's CMake builds them (add_library(degensac ...),
# add_library(matutls ...)); the shared object takes every degensac member and, as a static link
# would, only the matutls members that are referenced (ortho.c/unitary.c need CCMATH's unfl(),
# which the reference does not ship and never pulls in).
$(OUT)/libdegensac.a: $(DEG_OBJS)
	ar rcs $@ $(DEG_OBJS)
$(OUT)/libmatutls.a: $(MAT_OBJS)
	ar rcs $@ $(MAT_OBJS)

$(OUT)/libdegensac_ref.so: $(OUT)/libdegensac.a $(OUT)/libmatutls.a ref_shim.c ref.mk
	$(CC) -O2 -fPIC -shared -o $@ ref_shim.c -Wl,--whole-archive $(OUT)/libdegensac.a -Wl,--no-whole-archive \
	  $(OUT)/libmatutls.a -Wl,--wrap=time -L$(MKLDIR) -Wl,-rpath,$(MKLDIR) -lmkl_rt -lm
