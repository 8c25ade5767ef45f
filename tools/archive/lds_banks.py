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


# ---- ds_read_b64_tr_b16 over the TN kernel's DMA image: 1 KiB pieces of 2 rows x 512 B, 64 B of padding per piece, 32-byte
# window XOR with the row parity.  Lane (fi, fg) of a fragment read addresses row 4 fg + (fi >> 2) (+16 for the second read),
# bytes (fi & 3) * 8 of the fragment's 32-byte column block.  Lane groups of 32 (the guide's b64 grouping): each group must
# touch 64 distinct banks (8 rows x 8 banks).
def tn_addr(row, byte_in_row):
    return (row >> 1) * 1088 + (row & 1) * 512 + (byte_in_row ^ (32 * (row & 1)))


def check_tr(name, frag_byte):
    worst = 0
    for half in (0, 1):
        for grp in (range(0, 32), range(32, 64)):
            banks = {}
            for l in grp:
                fi, fg = l & 15, l >> 4
                a = tn_addr(16 * half + 4 * fg + (fi >> 2), frag_byte + (fi & 3) * 8)
                for b in ((a // 4) % 64, (a // 4 + 1) % 64):
                    banks.setdefault(b, set()).add(a // 4 * 4 if b == (a // 4) % 64 else a // 4 * 4 + 4)
            worst = max(worst, max(len(v) for v in banks.values()))
    print(name, "worst way:", worst)


for fb in (0, 32, 64, 96, 224, 256, 480):
    check_tr(f"TN tr-read frag byte {fb}", fb)
