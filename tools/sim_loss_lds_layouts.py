"""Bank-conflict cycles of the SpectralLoss forward kernel's LDS accesses (float2 elements) under candidate layouts of its
transform array (round 5; the model ranks the layouts the way the MI355X does - profiles/r05h_loss_ablation_b128.txt, addendum - but
what it predicts as 12 % fewer LDS cycles measured no faster: the shipped padding stays).
Model (MI355X guide): ds_read_b64: two groups of 32 lanes, 64 banks x 4 B (an 8-byte access takes a bank PAIR = one of 32 slots mod 32);
cycles of a group = max multiplicity of distinct addresses on a slot.  ds_write_b64: four groups of 16 lanes (array cycles), same slots."""
import numpy as np, itertools, sys

def plan(H):
    L=H.bit_length()-1
    N4 = 0 if L%3==0 else (1 if L%3==2 else 2)
    N8 = (L-2*N4)//3
    M = 1<<(2*N4)
    return L,N4,N8,M

def sl_pos(H,k):
    L,N4,N8,M=plan(H); p=0; kk=k; m=H
    for _ in range(N8): p+=(kk&7)*(m//8); kk>>=3; m>>=3
    for _ in range(N4): p+=(kk&3)*(m//4); kk>>=2; m>>=2
    return p

def patterns(H):
    """yields (kind, [address lists per wave-instruction]) for one block of 4096 points, 512 threads"""
    G=2048//H
    L,N4,N8,M=plan(H)
    out=[]
    T=np.arange(512)
    # radix-8 stages: q = H/8, .. down to >= M and > 1, plus unity if M == 1
    qs=[]
    q=H//8
    while q>=M and q>1: qs.append(q); q//=8
    if N8>0 and M==1: qs.append(1)
    first=True
    for q in qs:
        g=T//(H//8); r=T%(H//8); pos=r&(q-1) if q>1 else 0*r
        i0=g*H+((r-pos)<<3)+pos
        for m in range(8):
            if not first: out.append(('r',i0+m*q))        # the first stage is fused with the load: no reads
            out.append(('w',i0+m*q))
        first=False
    # radix-4 stages: q = M/4 .. 1 ; thread t takes butterflies 2t, 2t+1 ; n = 2G*H/4 = 1024 butterflies
    q=M//4
    while q>=1 and N4>0:
        for u in range(2):
            t=2*T+u
            g=t//(H//4); r=t%(H//4); pos=r&(q-1) if q>1 else 0*r
            i0=g*H+((r-pos)<<2)+pos
            for m in range(4):
                out.append(('r',i0+m*q)); out.append(('w',i0+m*q))
        q//=4
    # bins: e = tid + 512 trip ; g = e >> (log2H - 1), k = e & (H/2 - 1) ; reads base+ia, base+ib for sig 0,1
    for trip in range(2):
        e=T+512*trip
        g=e>>(L-1); k=e&(H//2-1)
        ia=np.array([sl_pos(H,int(x)) for x in k]); ib=np.array([sl_pos(H,int((H-x)&(H-1))) for x in k])
        for sig in range(2):
            base=(g+sig*G)<<L
            out.append(('r',base+ia)); out.append(('r',base+ib))
    return out

def cycles(layout, kind, addr):
    slots=layout(addr)
    tot=0
    grp=32 if kind=='r' else 16
    for w in range(0,512,64):
        for g0 in range(w,w+64,grp):
            s=slots[g0:g0+grp]; a=addr[g0:g0+grp]
            # multiplicity of distinct addresses per slot
            d={}
            mod = 32 if kind=='r' else 16      # ds_write_b64: 32 banks of 4 B = 16 float2 slots
            for sl,ad in zip(s%mod,a): d.setdefault(int(sl),set()).add(int(ad))
            tot+=max(len(v) for v in d.values())
    return tot

def evaluate(layout):
    res={}
    total=0; ideal=0
    for H in (1024,512,256,128,64,32):
        c=0; idl=0
        for kind,addr in patterns(H):
            c+=cycles(layout,kind,addr); idl+=(16 if kind=='r' else 32)
        res[H]=(c,idl); total+=c; ideal+=idl
    return total,ideal,res

layouts={
 'shipped i+2(i>>4)': lambda i: i+2*(i>>4),
 'none': lambda i: i,
 'i+(i>>5)': lambda i: i+(i>>5),
 'i+(i>>4)': lambda i: i+(i>>4),
 'i+(i>>3)': lambda i: i+(i>>3),
 'i+2(i>>5)': lambda i: i+2*(i>>5),
 'i+4(i>>5)': lambda i: i+4*(i>>5),
 'i+(i>>4)+(i>>8)': lambda i: i+(i>>4)+(i>>8),
 'i+2(i>>4)+(i>>7)': lambda i: i+2*(i>>4)+(i>>7),
 'i+(i>>5)+(i>>8)': lambda i: i+(i>>5)+(i>>8),
 'i+3(i>>5)': lambda i: i+3*(i>>5),
 'i+(i>>2)': lambda i: i+(i>>2),
 'xor (i>>5)&7<<2': lambda i: i^(((i>>5)&7)<<2),
 'xor (i>>5)&31': lambda i: i^((i>>5)&31),
 'xor (i>>3)&31 ... rot': lambda i: (i&~31)|((i+(i>>5))&31),
 'rot2: low5 + 2(i>>5)': lambda i: (i&~31)|((i+2*(i>>5))&31),
 'rot5: low5 + 5(i>>5)': lambda i: (i&~31)|((i+5*(i>>5))&31),
 'rot9': lambda i: (i&~31)|((i+9*(i>>5))&31),
 'rot(i>>3)': lambda i: (i&~31)|((i+(i>>3))&31),
 'rot(i>>4)*2': lambda i: (i&~31)|((i+2*(i>>4))&31),
}
if __name__=='__main__':
    for name,f in layouts.items():
        t,idl,res=evaluate(f)
        print('%-28s total %6d  ideal %6d  x%.3f   per H: %s'%(name,t,idl,t/idl,' '.join('%d:%.2f'%(H,c/i) for H,(c,i) in res.items())))
