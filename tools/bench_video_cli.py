import os, sys, contextlib
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"color-transfer_amd")]
from utils import cli
for g in (4, 8, 4, 8):
    t={}
    with contextlib.redirect_stdout(sys.stderr):
        cli.main(["test","--config",os.path.join(ROOT,"color-transfer_amd","configs","others.yaml"),"--model.metrics","psnr","--data.data_dir","null","--data.synthetic","video_u8","--data.n_frames","1000","--data.height","1080","--data.width","1920","--data.group",str(g)], timing=t)
    print(g, 1000/t["seconds"], t["h2d_bytes"]/t["seconds"]/1e9)
