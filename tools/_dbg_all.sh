for so in warpsense_amd/variants/*.so; do echo $so; WS_HIP_LIB=$PWD/$so python tools/_dbg1.py 2>&1 | grep differ; done
echo main; python tools/_dbg1.py 2>&1 | grep differ
