R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_unet3d.py -m gpu -q -x 2>&1 | tail -3
python tools/unet_profile.py 64 256 256 2>&1 | grep -v amdgpu | head -12
python tools/unet_profile.py 32 128 128 2>&1 | grep -v amdgpu | head -8
