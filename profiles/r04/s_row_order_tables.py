import itertools, collections
W=10
GROUPS=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
ORDER=[0,1,2,3,12,13,14,15,4,5,6,7,8,9,10,11]   # subset A first, then subset B
A=set([0,1,2,3,12,13,14,15])
def ent(b,y,x,S): return 24+b*S+(y+1)*W+x
def need(f,t,nb):
    if nb==1: return True
    return {6:t//3!=0,7:t//3!=2,8:t%3!=0,9:t%3!=2}.get(f,True)
def zero_p(S,nb):
    z=[11,12]
    if nb==2: z+=list(range(24+100+11, 24+S+9-11+1))  # p - 11 >= 124 (b0's y = 9 row), p + 11 <= 24 + S + 9 (end of b1's y = -1 row)
    return z
def pads(l,S,nb):
    """zero entries for padding of subset A and subset B, or None"""
    used={ent(*s,S)%16 for s in l}
    n=len(l); needA = n<8; needB = n<16
    pick={}
    for sub,needed in (('B',needB),('A',needA)):
        if not needed: continue
        for p in zero_p(S,nb):
            if p%16 not in used:
                used.add(p%16); pick[sub]=p; break
        else: return None
    return pick
def build2(S):
    res=lambda s: ent(*s,S)%16
    corners=[(b,y,x) for b in range(2) for y in (0,8) for x in (0,8)]
    best=None
    for assign in itertools.product(range(3), repeat=8):
        frs={6:[],7:[],8:[],9:[]}; rest=[]
        for b in range(2):
            for k in range(1,8):
                frs[6].append((b,0,k)); frs[7].append((b,8,k)); frs[8].append((b,k,0)); frs[9].append((b,k,8))
        for c,a in zip(corners,assign):
            b,y,x=c
            rowf=6 if y==0 else 7; colf=8 if x==0 else 9
            if a==0: frs[rowf].append(c)
            elif a==1: frs[colf].append(c)
            else: rest.append(c)
        out={}; ok=True
        for f,l in frs.items():
            seen=set(); keep=[]
            for s in sorted(l):
                if res(s) in seen or len(keep)==16: rest.append(s)
                else: seen.add(res(s)); keep.append(s)
            if pads(keep,S,2) is None and len(keep)<16:
                # drop one more square to free an available residue
                done=False
                for i in range(len(keep)):
                    k2=keep[:i]+keep[i+1:]
                    if pads(k2,S,2) is not None: rest.append(keep[i]); keep=k2; done=True; break
                if not done: ok=False
            out[f]=keep
        if not ok: continue
        pool=[(b,y,x) for b in range(2) for y in range(1,8) for x in range(1,8)]+rest
        cnt=collections.Counter(res(s) for s in pool)
        if max(cnt.values())>7: continue
        if best is None or len(rest)<best[0]: best=(len(rest),out,sorted(pool))
    if best is None: return None
    _,out,pool=best
    allt=[0,1,2,3,4,5,10]
    for f in allt: out[f]=[]
    for s in pool:
        for f in allt:
            if res(s) not in {res(q) for q in out[f]}:
                out[f].append(s); break
        else: raise SystemExit("no fit")
    return out
def build1():
    S=110; res=lambda s: ent(*s,S)%16
    out={f:[] for f in range(6)}
    for s in [(0,y,x) for y in range(9) for x in range(9)]:
        for f in range(6):
            if res(s) not in {res(q) for q in out[f]}:
                out[f].append(s); break
        else: raise SystemExit("no fit1")
    return out
def place(l,S,nb):
    rows=[None]*16
    for i,s in enumerate(l): rows[ORDER[i]]=('sq',ent(*s,S))
    pk=pads(l,S,nb)
    for p in range(16):
        if rows[p] is None: rows[p]=('pad',pk['A' if p in A else 'B'])
    return rows
def placem(l):
    rows=[-1]*16
    for i,(b,y,x) in enumerate(l): rows[ORDER[i]]=b*81+y*9+x
    return rows
def check(out,S,nb):
    seen=set()
    zero=set(range(24))
    for b in range(nb):
        base=24+b*S
        zero|=set(range(base,base+10))|set(range(base+100,base+110))|{base+10*(y+1)+9 for y in range(9)}
        if b+1<nb: zero|=set(range(base+110,base+S))
    for f,l in out.items():
        assert len(l)<=16
        for s in l:
            assert s not in seen; seen.add(s)
            b,y,x=s
            for t in range(9):
                if not need(f,t,nb):
                    dy,dx=t//3-1,t%3-1
                    assert not (0<=y+dy<9 and 0<=x+dx<9), (f,s,t)
        rows=place(l,S,nb)
        for t in range(9):
            off=(t//3-1)*10+(t%3-1)
            for kind,e in rows:
                if kind=='pad': assert e+off in zero, (f,e,off)
            for g in GROUPS:
                banks=collections.defaultdict(set)
                for lane in g:
                    e=rows[lane%16][1]+off
                    banks[e%16].add((lane//16,e))
                assert max(len(v) for v in banks.values())==1,(f,t)
    assert len(seen)==81*nb
def emit(out,nf):
    for f in range(nf):
        print("        {"+", ".join("%3d"%v for v in placem(out[f]))+"},")
o2=build2(119); check(o2,119,2); print("two boards, stride 119", zero_p(119,2)); emit(o2,11)
print(sum(1 for f in range(11) for t in range(9) if need(f,t,2)))
for f in range(11): print(f,len(o2[f]),pads(o2[f],119,2))
o1=build1(); check(o1,110,1); print("one board"); emit(o1,6)
for f in range(6): print(f,len(o1[f]),pads(o1[f],110,1))
