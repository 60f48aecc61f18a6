import csv,glob,sys
rows=[]
for f in glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]))
rows.sort()
ends=[i for i,r in enumerate(rows) if "grad_sqnorm" in r[2]]
pairs=[(a+1,b+1) for a,b in zip(ends,ends[1:]) if b-a>50]
lo,hi=pairs[-1]
seq=rows[lo:hi]
def find(s): return [i for i,r in enumerate(seq) if s in r[2]]
t0=seq[0][0]
marks=[("stem_fwd",(find("stem_fwd_mfma_k") or find("stem_fwd_k"))[0]),("avgpool_fwd",find("avgpool_fwd_k")[0]),("heads_fwd",find("heads_fwd_k")[0]),("heads_bwd_sample",find("heads_bwd_sample_k")[0]),("avgpool_bwd",find("avgpool_bwd_k")[0]),("stem_bwd",(find("stem_wgrad_mfma_k") or find("stem_bwd_weight"))[0])]
for n,i in marks: print(f"{n:18s} start {(seq[i][0]-t0)/1e3:9.1f} us  end {(seq[i][1]-t0)/1e3:9.1f}")
print("step span", (seq[-1][1]-t0)/1e3, "launches", len(seq), "prev step end->this start gap", (t0-rows[lo-1][1])/1e3)
busy=0; cur_s,cur_e=seq[0][0],seq[0][1]
for s,e,_ in seq[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("busy (union) us", busy/1e3, "sum durations", sum(e-s for s,e,_ in seq)/1e3)
