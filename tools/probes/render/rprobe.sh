for th in 32 96 160 480; do echo "TH=$th"; MIR_RENDER_TH=$th python3 tools/render_time.py 1024 2>&1 | grep "640x480"; done
for d in 1 3; do echo "DBG=$d"; MIR_RENDER_DBG=$d python3 tools/render_time.py 1024 2>&1 | grep "640x480"; done
echo "DBG=3 TH=480"; MIR_RENDER_DBG=3 MIR_RENDER_TH=480 python3 tools/render_time.py 1024 2>&1 | grep "640x480"
