#!/bin/bash
# tools/pkfma_probe.hip alone, then beside the other rank's workload (tools/diverge_probe.py stress role, 4-layer model).
cd "${GRAFT_REPO_ROOT:-.}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o tools/_pkfma_probe tools/pkfma_probe.hip || exit 1
echo "alone:"; tools/_pkfma_probe ${1:-8}
rm -f /tmp/pk_ready /tmp/pk_stop
python tools/diverge_probe.py --role stress --stress model --layers 4 --new 32 --ready /tmp/pk_ready --stop /tmp/pk_stop > /dev/null 2>&1 &
SP=$!
for i in $(seq 1 600); do [ -f /tmp/pk_ready ] && break; sleep 0.5; done
echo "beside the model workload of another process:"; tools/_pkfma_probe ${2:-20}
touch /tmp/pk_stop; wait $SP
