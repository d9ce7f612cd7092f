import numpy as np, sys
a=np.load(sys.argv[1]); us=lambda x: x/100.0
start,exp,com,ver,end,swe,fl=a.T[:7]
adopted=(fl&2)!=0
print("last committer commit %.0f  max end %.0f" % (us(com[adopted].max()), us(end.max())), " heavy bulks:", " ".join(f"{k}:{us(end[k]-swe[k]):.0f}" for k in (286,331,366,439,440,452)))
