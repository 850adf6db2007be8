# Build recipe for the REAL reference (aquaskyline/SOAP3-dp) CPU path, used only as
# test infrastructure (parity oracle + optional CPU baseline).  Sources are compiled
# where they lie under $(REF); nothing is copied into this repo.  Outputs go only to
# oracle/_ref/ (git-ignored, but shipped to the GPU box with the snapshot).
#
#   make -f oracle/ref.mk            # from the repo root
#
# Accommodations (documented in DESIGN.md):
#  * 2bwt-lib/BWT.c:424 compares a pointer with `> 0` inside BWTPrintMemoryUsage (a
#    debug print, not on the path) which g++ >= 11 rejects; the file is streamed through
#    sed into the compiler (no copy is written anywhere).
#  * SAM.cpp and 2bwt-flex/LTConstruct.c have non-void functions without `return`;
#    they are compiled at -O0 so the fall-through is harmless, as the reference authors'
#    toolchain treated it.
#  * The CUDA translation units (*.cu) are NOT built: no nvcc here and no stand-in CUDA
#    headers are written.  main()/alignment.cu are replaced by oracle/ref_harness.cpp,
#    which is our own driver calling the reference's host functions.

REF    ?= /root/reference
OUT    := oracle/_ref
OBJ    := $(OUT)/obj
CXX    := g++
CC     := gcc
FLAGS  := -O2 -fpermissive -w -mpopcnt -fopenmp
SAMFLG := -O2 -w -D_FILE_OFFSET_BITS=64 -D_LARGEFILE64_SOURCE -D_USE_KNETFILE -D_CURSES_LIB=0

BWTLIB := BWT dictionary DNACount HSP HSPstatistic iniparser inistrlib karlin MemManager \
          MiscUtilities QSufSort r250 TextConverter Timing Socket BWTConstruct
FLEX   := HOCC LT HOCCConstruct SRA2BWTCheckAndExtend SRA2BWTMdl LTConstruct
SAMT   := sam bam bgzf kstring bam_import faidx bam_pileup bam_aux sam_header razf knetfile
HOST   := BGS-IO BGS-HostAlgnmtAlgo2 BGS-HostAlgnmtAlgoSingle PE SAList CPUfunctions PEAlgnmt \
          OutputDPResult AlgnResult global_arrays aio_thread ssse3_popcount IniParam \
          UsageInterface FileUtilities IndexHandler QueryParser SAM

BWTOBJ  := $(addprefix $(OBJ)/lib_,$(addsuffix .o,$(BWTLIB)))
FLEXOBJ := $(addprefix $(OBJ)/flex_,$(addsuffix .o,$(FLEX)))
SAMOBJ  := $(addprefix $(OBJ)/sam_,$(addsuffix .o,$(SAMT)))
HOSTOBJ := $(addprefix $(OBJ)/host_,$(addsuffix .o,$(HOST)))

all: $(OUT)/soap3-dp-builder $(OUT)/BGS-Build $(OUT)/ref_harness $(OUT)/libdpref.so $(OUT)/ref_bam2sam $(OUT)/ref_sam2bam

$(OBJ):
	mkdir -p $(OBJ)

$(OBJ)/lib_BWT.o: $(REF)/2bwt-lib/BWT.c | $(OBJ)
	sed '424s/cachedSaIndex > 0/cachedSaIndex != 0/' $< | $(CXX) $(FLAGS) -x c++ -I$(REF)/2bwt-lib -c - -o $@

$(OBJ)/lib_%.o: $(REF)/2bwt-lib/%.c | $(OBJ)
	$(CXX) $(FLAGS) -x c++ -c $< -o $@

$(OBJ)/flex_LTConstruct.o: $(REF)/2bwt-flex/LTConstruct.c | $(OBJ)
	$(CXX) -O0 -fpermissive -w -x c++ -c $< -o $@

$(OBJ)/flex_%.o: $(REF)/2bwt-flex/%.c | $(OBJ)
	$(CXX) $(FLAGS) -x c++ -c $< -o $@

$(OBJ)/sam_%.o: $(REF)/samtools-0.1.18/%.c | $(OBJ)
	$(CC) $(SAMFLG) -c $< -o $@

$(OBJ)/host_SAM.o: $(REF)/SAM.cpp | $(OBJ)
	$(CXX) -O0 -fpermissive -w -mpopcnt -c $< -o $@

$(OBJ)/host_%.o: $(REF)/%.cpp | $(OBJ)
	$(CXX) $(FLAGS) -c $< -o $@

# the reference's own index tools (their own main()s)
$(OUT)/soap3-dp-builder: $(REF)/2bwt-flex/2BWT-Builder.c $(BWTOBJ) $(FLEXOBJ)
	$(CXX) $(FLAGS) -x c++ $< -x none $(BWTOBJ) $(FLEXOBJ) -lm -o $@
	cp oracle/builder.ini $(OUT)/soap3-dp-builder.ini

$(OUT)/BGS-Build: $(REF)/BGS-Build.cpp $(BWTOBJ)
	$(CXX) $(FLAGS) $< $(BWTOBJ) -lm -o $@

# The reference's DP ENGINES: the host halves of DV-DPfunctions.cu (lines 742-3911: packers, decode, CIGAR encoder, seeding, candidate merges,
# output filters, the thread plumbing) and their callers DV-SemiDP.cu / DV-DPForBothUnalign.cu / DV-DPForSingleReads.cu are plain C++ once the CUDA
# runtime calls are out of the way.  They are streamed through sed into g++ from where they lie; the edits, all deletions of CUDA-only lines
# except one, are:
#   DV-DPfunctions.h   `#include <cuda.h>`, `static cudaError_t gpuErr;`, `cudaEvent_t start, stop;` (TimeRecorder, unused), the five
#                      `CUcontext ctx;` members, the four `cudaFree ( _x );` lines of SOAP3Wrapper::freeIndex  -- written to obj/dpinc/ (never into the
#                      repository), because the .cu files include it by name
#   alignment.h        `#include "DV-Kernel.h"` (kernel prototypes with __global__ / __constant__)                 -- obj/dpinc/ likewise
#   DV-DPfunctions.cu  lines 1-741 are not used (includes, the device code -- built for gfx950 into libdpref.so below -- and SemiGlobalAligner's
#                      cudaMalloc / cudaMemcpy / launch code, whose contract oracle/ref_dp_glue.cpp implements on host memory); the 22
#                      `cuCtxPushCurrent / cuCtxPopCurrent` lines (context hand-over between host threads); and line 2326, which names the type
#                      through the constructor (`HalfEndAlignmentEngine::HalfEndAlignmentEngine *`): g++ >= 5 rejects that, the class name is used
#   DV-DPForSingleReads.cu  `#include <cuda.h>`
# No stand-in headers or libraries are written.
DPINC := $(OBJ)/dpinc
$(DPINC)/DV-DPfunctions.h: $(REF)/DV-DPfunctions.h | $(OBJ)
	mkdir -p $(DPINC)
	sed -e '/#include <cuda.h>/d' -e '/^static cudaError_t gpuErr;/d' -e '/cudaEvent_t start, stop;/d' -e '/^ *CUcontext ctx;/d' -e '/^ *cudaFree ( _[a-zA-Z]* );$$/d' $< > $@
$(DPINC)/alignment.h: $(REF)/alignment.h | $(OBJ)
	mkdir -p $(DPINC)
	sed -e '/#include "DV-Kernel.h"/d' $< > $@
DPHDR := $(DPINC)/DV-DPfunctions.h $(DPINC)/alignment.h
$(OBJ)/dp_host.o: $(REF)/DV-DPfunctions.cu $(DPHDR)
	( echo '#include "DV-DPfunctions.h"'; echo '#include "OutputDPResult.h"'; echo '#include <assert.h>'; echo '#include <algorithm>'; echo 'using namespace std;'; \
	  sed -n '742,3911p' $< | sed -e '/cuCtxPopCurrent\|cuCtxPushCurrent/d' \
	      -e 's/^HalfEndAlignmentEngine::HalfEndAlignmentEngine \* HalfEndAlignmentEngine::engine;/HalfEndAlignmentEngine * HalfEndAlignmentEngine::engine;/' ) | \
	$(CXX) $(FLAGS) -I$(DPINC) -I$(REF) -x c++ -c - -o $@
$(OBJ)/dp_%.o: $(REF)/DV-%.cu $(DPHDR)
	sed -e '/#include <cuda.h>/d' $< | $(CXX) $(FLAGS) -I$(DPINC) -I$(REF) -x c++ -c - -o $@
DPOBJ := $(OBJ)/dp_host.o $(OBJ)/dp_SemiDP.o $(OBJ)/dp_DPForBothUnalign.o $(OBJ)/dp_DPForSingleReads.o
$(OBJ)/glue.o: oracle/ref_dp_glue.cpp $(DPHDR)
	$(CXX) $(FLAGS) -I$(DPINC) -I$(REF) -c $< -o $@
$(OBJ)/oracle_dp.o: oracle/s3_oracle_dp.c | $(OBJ)
	$(CC) -O2 -c $< -o $@

# our driver around the reference's CPU alignment path
$(OUT)/ref_harness: oracle/ref_harness.cpp $(HOSTOBJ) $(BWTOBJ) $(FLEXOBJ) $(SAMOBJ) $(DPOBJ) $(OBJ)/glue.o $(OBJ)/oracle_dp.o
	$(CXX) $(FLAGS) -I$(REF) $< $(DPOBJ) $(OBJ)/glue.o $(OBJ)/oracle_dp.o $(HOSTOBJ) $(filter-out $(OBJ)/lib_BWTConstruct.o,$(BWTOBJ)) $(FLEXOBJ) $(SAMOBJ) -lz -lm -lpthread -o $@

# The reference's semi-global DP DEVICE code (DV-DPfunctions.cu:35-512) built for gfx950 with hipcc -- CUDA and HIP share the kernel language.
# The region is streamed out of the reference file (never copied into the repository; obj/ does not travel to the GPU box either), with
# the two edits CDNA forces: no texture units, so the `texture <uint>` declarations go and MC_ReadUnpack indexes its array argument (the
# texture was bound to exactly that array, DV-DPfunctions.cu:660).  Four lines of DV-DPfunctions.h (its typedefs / macros; the header itself
# includes <cuda.h>) are streamed the same way.  The host driver around it is ours: oracle/ref_dp_driver.hip.
HIPCC ?= /opt/rocm/bin/hipcc
$(OBJ)/dp_device.inc: $(REF)/DV-DPfunctions.cu | $(OBJ)
	sed -n '35,512p' $< | sed -e '/^texture <uint> tex/d' -e 's/tex1Dfetch(texPatterns, readTPARA + (((i)>>4)<<5))/X[readTPARA + (((i)>>4)<<5)]/' > $@
	! grep -q "tex1Dfetch\|^texture" $@
$(OBJ)/dp_header_lines.inc: $(REF)/DV-DPfunctions.h | $(OBJ)
	sed -n -e '/^#define DP_THREADS_PER_BLOCK/p' -e '/^typedef unsigned char uchar;/p' -e '/^typedef unsigned int uint;/p' -e '/^#define MC_CeilDivide16/p' $< > $@
$(OUT)/libdpref.so: oracle/ref_dp_driver.hip $(OBJ)/dp_device.inc $(OBJ)/dp_header_lines.inc
	$(HIPCC) -O2 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -Ioracle -o $@ $<

# BAM reader for the tests: the reference's samtools decodes what `soap3-dp -bam` wrote (oracle/ref_bam2sam.c is ours)
$(OUT)/ref_bam2sam: oracle/ref_bam2sam.c $(SAMOBJ)
	$(CC) $(SAMFLG) -I$(REF)/samtools-0.1.18 $< $(SAMOBJ) -lz -lm -o $@

# BAM writer for the fixtures of the `-bam` input (oracle/ref_sam2bam.c is ours; the library is the reference's)
$(OUT)/ref_sam2bam: oracle/ref_sam2bam.c $(SAMOBJ)
	$(CC) $(SAMFLG) -I$(REF)/samtools-0.1.18 $< $(SAMOBJ) -lz -lm -o $@

clean:
	rm -rf $(OUT)

.PHONY: all clean
