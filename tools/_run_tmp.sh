cp s2p_amd/lib/libs2p_hip.so /tmp/orig.so
for v in old cur; do cp build/variants/$v/libs2p_hip.so s2p_amd/lib/libs2p_hip.so; echo "== $v"; python tools/mgm_multi_stages.py 1000 256 2>&1 | grep -v amdgpu; python tools/mgm_multi_stages.py 697 192 2>&1 | grep -v amdgpu; done
cp /tmp/orig.so s2p_amd/lib/libs2p_hip.so
