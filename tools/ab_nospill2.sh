#!/bin/bash
# the final no-spill policy (ns4) against the spilling build (ns0) on the shapes whose builds changed
source tools/ab_quick.sh "--kernel 17|--kernel 31|--model gain-blk-offset --kernel 9|--model gain-blk-offset --kernel 11|--model gain --kernel 15 --nodata 2|--model gain --kernel 13 --nodata 2|--config 3 --no-end-to-end" _ab/lib_ns0.so _ab/lib_ns4.so
