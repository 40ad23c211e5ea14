set -u
R=$PWD; L=$R/tools/probe/ab/libabl.so
for r in 1 0; do for a in 0 2 8 10 4 14; do echo "C64R=$r ABLATE=$a: $(VPD_C64R=$r VPD_LIB_PATH=$L VPD_ABLATE=$a python3 tools/bench_conv.py 256 2>/dev/null | grep layer1)"; done; done
