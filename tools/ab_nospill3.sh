#!/bin/bash
source tools/ab_quick.sh "--kernel 17|--kernel 31|--kernel 21" _ab/lib_ns0.so _ab/lib_ns6.so
