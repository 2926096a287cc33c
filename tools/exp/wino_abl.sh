# timing-only ablations of the Winograd kernel (results are wrong for ABL != 0); needs a development build of the library:
#   DVIS_HIPCC_FLAGS=-DDVIS_WINO_ABLATION python -m dvis_plus_amd.build --force
for abl in 0 1 2 3 4 7 8 11; do echo -n "ABL=$abl  "; DVIS_WINO_ABL=$abl python tools/winograd_time.py 2>&1 | grep "FPN\|res3" | awk '{for(i=1;i<=NF;i++) if($i=="own" && $(i+2)=="us") printf "%s %s us   ", $1, $(i+1)}'; echo; done
