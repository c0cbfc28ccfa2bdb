#!/bin/bash
# round 5: the layer-edge lab (one all-to-all edge inside a launch vs two launches)
mkdir -p gpurun_out
timeout 120 tools/bin/layer_edge_lab > gpurun_out/layer_edge_lab.log 2>&1
echo "exit $?" >> gpurun_out/layer_edge_lab.log
cat gpurun_out/layer_edge_lab.log
