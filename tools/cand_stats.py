import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, oracle_lib as oracle
from orb_slam2_aruco_amd import synth
import collections
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 640)
img = synth.stream(rows, cols, 2, 1000, "ARUCO", n_markers=4)[1]
ora = oracle.ArucoOracle("ARUCO"); ora.detect(img)
th = ora.stage_image(0)
fg = np.zeros((rows+2, cols+2), np.uint8); fg[1:-1,1:-1] = (th != 0)
DX=[1,1,0,-1,-1,-1,0,1]; DY=[0,-1,-1,-1,0,1,1,1]
def ring(x,y): return [fg[y+DY[d], x+DX[d]] for d in range(8)]
stats = collections.Counter(); steps = collections.Counter()
ncand=0
for y in range(1, rows+1):
    if y % 32 == 0: continue
    for x in range(1, cols+1):
        for hole in (0,1):
            if not hole:
                if not (fg[y,x] and not fg[y,x-1] and not fg[y-1,x-1] and not fg[y-1,x] and not fg[y-1,x+1]): continue
                sx=x
            else:
                if not (not fg[y,x] and fg[y,x-1] and fg[y-1,x]): continue
                sx=x-1
            ncand+=1
            r=ring(sx,y)
            if not any(r): stats['single']+=1; continue
            d0 = 0 if hole else 4
            s=None
            for j in range(7,-1,-1):
                if r[(d0+j)&7]: s=(d0+j)&7; break
            s0=s; cx,cy=sx,y; n=0; key0=y*65536+x
            fate=None
            while True:
                if cx%32==0 or cy%32==0: fate='grid'; break
                if not hole and cy*65536+cx < key0: fate='noncanon'; break
                if hole and cy*65536+cx+1 < key0 - 65536*0 and cy < y: fate='noncanon'; break
                r=ring(cx,cy)
                for i in range(1,9):
                    d=(s+i)&7
                    if r[d]: break
                cx+=DX[d]; cy+=DY[d]; s=(d+4)&7; n+=1
                if cx==sx and cy==y and s==s0: fate='closed>70' if n>70 else 'closed'; break
                if n>2000: fate='runaway'; break
            stats[fate]+=1; steps[fate]+=n
print(rows, cols, "candidates", ncand, dict(stats)); print("steps", dict(steps), "total", sum(steps.values()))
# segment steps: total border states on gridded borders ~ count border pixels
b = oracle.find_contours(th); print("borders", len(b), "points", sum(len(x) for x in b), "kept", sum(len(x)>70 for x in b))
