#!/bin/bash
# usage: scripts/run_overlap_stages.sh <Data directory> [alignMethod]
# The overlap-detection stages of `shasta --config Nanopore-May2022` (conf/Nanopore-May2022.conf, BASELINE
# configs[3]) on an existing Data/ directory, in the order of srcMain/main.cpp:640-722, with the stage executable
# of this repository (shasta_amd/_build/shasta_mi355x_stage; needs an MI355X).  Data/ must hold what the
# reference's read loading and k-mer selection wrote: Reads-*, ReadNames, ReadMetaData, ReadFlags, Kmers.
# alignMethod: 3 (the configuration's own; default) or 4 (the method BASELINE's metric is quoted on).
set -e
DATA=${1:?Data directory}
METHOD=${2:-3}
STAGE=$(dirname "$0")/../shasta_amd/_build/shasta_mi355x_stage
$STAGE markers      "$DATA"                                   # Assembler::findMarkers
$STAGE palindromic  "$DATA" 100 100 10 0.1 0.1 100            # Reads.palindromicReads.* defaults
$STAGE lowhash0     "$DATA" 4 0.01 10 20 0 5 30 5             # MinHash: m hashFraction iterations perRead log2Buckets minBucket maxBucket minFrequency
$STAGE suppress     "$DATA" 30                                # Align.sameChannelReadAlignment.suppressDeltaThreshold
$STAGE candidate-table "$DATA"
$STAGE align        "$DATA" 10 0.1 100 100 100 0 "$METHOD" 0.05 10    # minAlignedMarkerCount minAlignedFraction maxSkip maxDrift maxTrim suppressContainments alignMethod downsamplingFactor bandExtend
# ReadGraph.creationMethod 2 of the configuration is outside this repository; method 0 is:
# $STAGE read-graph "$DATA" 6 30
