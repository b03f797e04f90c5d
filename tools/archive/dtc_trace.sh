#!/bin/bash
# on the GPU box: the per-phase trace of the temporal kernels with a lab build (PCAA_DTC_TRACE)
set -e
PCAA_HIPCC_EXTRA=-DPCAA_DTC_TRACE python -m opensetgaitrecognition_pcaa_amd.build > /dev/null
PCAA_HIPCC_EXTRA=-DPCAA_DTC_TRACE python tools/dtc_lab.py --trace
