# ds_read_b128 conflict check per guide lane groups
G128 = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
        list(range(4,12))+list(range(16,20))+list(range(28,32)),
        list(range(32,36))+list(range(44,48))+list(range(52,60)),
        list(range(36,44))+list(range(48,52))+list(range(60,64))]
F=[0,2,3,1]
def check(addr_fn, name):
    worst=0
    for grp in G128:
        slots={}
        for l in grp:
            a=addr_fn(l)
            s=(a//16)%16
            slots.setdefault(s,set()).add(a)
        w=max(len(v) for v in slots.values())
        worst=max(worst,w)
    print(name,"worst way:",worst)
# X read (round 1): row = base + fi, pos = fg ^ F[(fi>>2)&3]
for base in (0,16,64,128):
    check(lambda l: (base+(l&15))*64 + (((l>>4) ^ F[((l&15)>>2)&3])*16), f"X base{base}")
# W read bf16 mapping: row = wn*64 + 8*(fi>>2)+(fi&3)+4*(fn&1)+32*(fn>>1); pos = fg ^ F[(row>>3)&3]
for wn in range(4):
  for fn in range(4):
    def ad(l):
        fi=l&15; fg=l>>4
        row=wn*64+8*(fi>>2)+(fi&3)+4*(fn&1)+32*(fn>>1)
        return row*64 + ((fg ^ F[(row>>3)&3])*16)
    check(ad, f"Wbf wn{wn} fn{fn}")
