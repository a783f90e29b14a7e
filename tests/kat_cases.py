"""Known-answer tests of the reference CPU baker, as data.

Each case names a texture generator of tests/native/kat_textures.c, the UV geometry, the bake
settings and the expected `ommDebugStats` counts asserted by
/root/reference/support/tests/test_omm_bake_cpu.cpp (line in `ref`).  The reference runs every
case in six suite configurations (`:2581-2589`): default, DisableZOrder, Force32BitIndices, both,
texture alphaCutoff (SAT / coarse pass on), and serialisation round trip -- CONFIGS below restates
the bake-relevant ones; the expected numbers are identical in all of them.
"""
import numpy as np

QUAD_UV = [0, 0, 0, 1, 1, 0, 1, 1]          # :594-595
QUAD_IX = [0, 1, 2, 3, 1, 2]
QUAD2_UV = [0, 0, 0, 1, 1, 1, 1, 0]         # :1395-1397
QUAD2_IX = [0, 1, 2, 1, 2, 3]
TRI_A = [0.2, 0.0, 0.1, 0.8, 0.9, 0.1]      # :1131
DEGEN_V = [0.2, 0.0, 0.2, 0.437582970, 0.2, 0.218791485]   # :2309-2311

# name -> (disable_zorder, force32, sat)
CONFIGS = {
    "Default": (False, False, False),
    "TextureDisableZOrder": (True, False, False),
    "Force32BitIndices": (False, True, False),
    "TextureAsUNORM8": (True, True, False),      # enum value 3 = both bits (:32-40,92-96)
    "AlphaCutoff": (False, False, True),
    "Serialize": (True, False, True),            # enum value 5; the blob round trip itself is next-tier
}


def hex_grid(n=32, m=32):
    """HexagonsReuse geometry, :1490-1509 (fp32 arithmetic as written there)."""
    uv, ix = [], []
    f = np.float32
    for j in range(m):
        for i in range(n):
            o = 3 * (i + j * n)
            ix += [o, o + 1, o + 2]
            ox, oy = f(i) / f(n), f(j) / f(m)
            for (a, b) in ((0., 0.), (0., 1.), (1., 1.)):
                uv += [ox + f(a) / f(n), oy + f(b) / f(m)]
    return np.array(uv, np.float32), np.array(ix, np.uint32)


def S(O=0, T=0, UT=0, UO=0, FO=0, FT=0, FUO=0, FUT=0):
    return dict(O=O, T=T, UT=UT, UO=UO, FO=FO, FT=FT, FUO=FUO, FUT=FUT)


def case(name, ref, tex, level, expect, uv=QUAD_UV, ix=QUAD_IX, size=(1024, 1024), param=0.0, slow=False, **opt):
    return dict(name=name, ref=ref, tex=tex, level=level, expect=expect, uv=uv, ix=ix, size=size, param=param, slow=slow, opt=opt)


CASES = [
    case("AllOpaque4", 791, "const", 4, S(FO=2), param=0.6),
    case("AllOpaque3", 804, "const", 3, S(FO=2), param=0.6),
    case("AllOpaque2", 816, "const", 2, S(FO=2), param=0.6),
    case("AllOpaque1", 828, "const", 1, S(FO=2), param=0.6),
    case("AllOpaque0", 840, "const", 0, S(FO=2), param=0.6),
    case("AllTransparent4", 852, "const", 4, S(FT=2), param=0.4),
    case("AllTransparent3", 864, "const", 3, S(FT=2), param=0.4),
    case("AllTransparent2", 876, "const", 2, S(FT=2), param=0.4),
    case("AllTransparent1", 888, "const", 1, S(FT=2), param=0.4),
    case("AllUnknownTransparent", 900, "diag8", 1, S(FUT=2), param=0.0),
    case("AllUnknownOpaque", 914, "diag8", 1, S(FUO=2), param=1.0),
    case("AllTransparentOpaqueCorner4", 928, "corner", 4, S(T=255, UT=1, FT=1)),
    case("Circle", 958, "circle", 4, S(O=204, T=219, UT=39, UO=50)),
    case("CircleMergeSimilar", 973, "circle", 4, S(O=200, T=216, UT=42, UO=54), merge_similar=True),
    case("CircleOC2", 988, "circle", 4, S(O=254, T=258), fmt=1),
    case("SineUNORM8", 1001, "sine_u8", 4, S(O=128, T=256, UT=48, UO=80)),
    case("Sine", 1021, "sine", 4, S(O=224, T=128, UT=96, UO=64)),
    case("SineOC2", 1043, "sine", 4, S(O=288, T=224), fmt=1),
    case("SineOC2Neg", 1063, "sine", 4, S(O=288, T=224), fmt=1),   # (the reference test body equals SineOC2's)
    case("Mandelbrot", 1083, "mandelbrot", 5, S(O=1212, T=484, UT=124, UO=228)),
    case("Mandelbrot2", 1124, "mandelbrot", 5, S(O=521, T=286, UT=82, UO=135), uv=TRI_A, ix=[0, 1, 2]),
    case("Mandelbrot3", 1169, "mandelbrot", 9, S(O=164040, T=91320, UT=3039, UO=3745), uv=TRI_A, ix=[0, 1, 2]),
    case("Julia", 1243, "julia", 9, S(O=254265, T=5055, UT=1336, UO=1488), uv=TRI_A, ix=[0, 1, 2]),
    case("Julia_UVFP16", 1266, "julia", 9, S(O=254321, T=5108, UT=1264, UO=1451), uv=TRI_A, ix=[0, 1, 2], uv_format="fp16"),
    case("Julia_UV_UNORM16", 1290, "julia", 9, S(O=254325, T=5110, UT=1284, UO=1425), uv=TRI_A, ix=[0, 1, 2], uv_format="unorm16"),
    case("JuliaUNORM8", 1314, "julia_u8", 9, S(O=254251, T=5176, UT=1215, UO=1502), uv=TRI_A, ix=[0, 1, 2]),
    case("Julia_T_AND_UO", 1337, "julia_u8", 9, S(O=0, T=5176, UT=1215, UO=1502 + 254251), uv=TRI_A, ix=[0, 1, 2], le=0, gt=3),
    case("Julia_FLIP_T_AND_O", 1363, "julia_u8", 9, S(O=5176, T=254251, UT=1502, UO=1215), uv=TRI_A, ix=[0, 1, 2], le=1, gt=0),
    case("Uniform", 1389, "uniform4", 6, S(O=5132, T=2393, UT=357, UO=310), uv=QUAD2_UV, ix=QUAD2_IX, size=(4, 4)),
    case("HexagonsLvl6", 1422, "hexagons", 6, S(O=902, T=0, UT=3, UO=7287), uv=QUAD2_UV, ix=QUAD2_IX),
    case("HexagonsLvl8", 1454, "hexagons", 8, S(O=77995, T=535, UT=23163, UO=29379), uv=QUAD2_UV, ix=QUAD2_IX),
    case("HexagonsReuseLvl2", 1486, "hexagons", 2, S(O=6933, UT=1935, UO=7516), uv="hexgrid", ix="hexgrid"),
    case("HexagonsReuseLvl3", 1532, "hexagons", 3, S(O=40134, T=250, UT=11939, UO=13213), uv="hexgrid", ix="hexgrid"),
    case("HexagonsReuseLvl4", 1579, "hexagons", 4, S(O=187129, T=17979, UT=30309, UO=26727), uv="hexgrid", ix="hexgrid"),
    case("HexagonsReuseLvl5", 1626, "hexagons", 5, S(O=796515, T=138195, UT=56743, UO=57123), uv="hexgrid", ix="hexgrid"),
    case("HexagonsReuseLSH", 1673, "hexagons", 4, S(O=170724, T=11380, UT=37864, UO=39104, FT=12), uv="hexgrid", ix="hexgrid", merge_similar=True),
    # degenerate (zero-area) triangles, :2306-2534
    case("Degen_Default_lvl1", 2306, "circle", 1, S(O=1, UT=1, UO=2), uv=DEGEN_V, ix=[0, 1, 2]),
    case("Degen_Default_lvl2", 2322, "circle", 2, S(O=6, T=3, UT=3, UO=4), uv=DEGEN_V, ix=[0, 1, 2]),
    case("Degen_Default_Horizontal", 2339, "circle", 1, S(O=0, T=3, UT=1), uv=[0.2, 0.2, 0.3, 0.2, 0.41, 0.2], ix=[0, 1, 2]),
    case("Degen_Default_Diagonal", 2355, "circle", 2, S(T=13, UT=2, UO=1), uv=[0.2, 0.2, 0.3, 0.2, 0.4, 0.2], ix=[0, 1, 2]),
    case("Degen_Default_lvl3", 2371, "circle", 3, S(O=28, T=21, UT=7, UO=8), uv=DEGEN_V, ix=[0, 1, 2]),
    case("Degen_Default_lvl4", 2388, "circle", 4, S(O=136, T=91, UT=14, UO=15), uv=DEGEN_V, ix=[0, 1, 2]),
    case("Degen_Default_lvl4_wrap", 2405, "circle", 4, S(O=136, T=91, UT=14, UO=15),
         uv=[-0.8, 0.0, -0.8, 0.437582970, -0.8, 0.218791485], ix=[0, 1, 2], addr=0),
    case("Degen_Default_dyn_lvl_0_1", 2423, "circle", 12, S(O=9642463, T=7108335, UT=3771, UO=22647), uv=DEGEN_V, ix=[0, 1, 2], dyn_scale=0.1, slow=True),
    case("Degen_Default_dyn_lvl_0_5", 2440, "circle", 12, S(O=601591, T=443211, UT=942, UO=2832), uv=DEGEN_V, ix=[0, 1, 2], dyn_scale=0.5),
    case("Degen_Default_dyn_lvl_2", 2457, "circle", 12, S(O=37333, T=27495, UT=353, UO=355), uv=DEGEN_V, ix=[0, 1, 2], dyn_scale=2.0),
    case("Degen_Default_dyn_lvl_3", 2474, "circle", 12, S(O=37333, T=27495, UT=353, UO=355), uv=DEGEN_V, ix=[0, 1, 2], dyn_scale=3.0),
    case("Degen_Default_dyn_lvl_10", 2491, "circle", 12, S(O=2266, T=1653, UT=87, UO=90), uv=DEGEN_V, ix=[0, 1, 2], dyn_scale=10.0),
    case("Degen_Point_Transparent", 2508, "circle", 12, S(FT=1), uv=[0.2, 0.437582970] * 3, ix=[0, 1, 2], dyn_scale=2.0),
    case("Degen_Point_Opaque", 2521, "circle", 12, S(FO=1), uv=[0.2, 0.1] * 3, ix=[0, 1, 2], dyn_scale=2.0),
    case("Invalid_FullyUnknownTransparent", 2536, "circle", 4, S(FUT=1),
         uv=[0.0, 0.0, 0.0, float("nan"), 0.0, 0.221271083], ix=[0, 1, 2], unresolved=-3),
]

# Leaflet cases (`:640-746`): fp32 texture = 1 - blue/255 of assets/tests/leaflet.png, optional box-filtered mips.
#   LeafletMipN(mipStart, numMip, alphaCutoff): level 6, UV (0.05,0.1),(0.1,0.9),(0.9,0.9)
#   LeafletLevelN(level): UV (0.35,0.1),(0.1,0.9),(0.9,0.8), special indices disabled
LEAFLET_MIP = [
    ("Leaflet_Alpha_0_2", 1721, 0, 1, 0.2, S(O=864, T=2712, UT=275, UO=245)),
    ("LeafletMip0_to_0", 1733, 0, 1, 0.5, S(O=817, T=2763, UT=232, UO=284)),
    ("LeafletMip0_to_1", 1745, 0, 2, 0.5, S(O=809, T=2720, UT=275, UO=292)),
    ("LeafletMip0_to_2", 1757, 0, 3, 0.5, S(O=784, T=2688, UT=307, UO=317)),
    ("LeafletMip0_to_3", 1769, 0, 4, 0.5, S(O=776, T=2684, UT=311, UO=325)),
    ("LeafletMip0_to_4", 1781, 0, 5, 0.5, S(O=724, T=2586, UT=409, UO=377)),
    ("LeafletMip0_to_5", 1793, 0, 6, 0.5, S(O=615, T=2430, UT=565, UO=486)),
    ("LeafletMip0_to_6", 1805, 0, 7, 0.5, S(O=349, T=2408, UT=587, UO=752)),
    ("LeafletMip0_to_7", 1817, 0, 8, 0.5, S(O=0, T=2408, UT=587, UO=1101)),
    ("LeafletMip1", 1841, 1, 1, 0.5, S(O=847, T=2728, UT=248, UO=273)),
    ("LeafletMip2", 1853, 2, 1, 0.5, S(O=857, T=2725, UT=268, UO=246)),
    ("LeafletMip3", 1865, 3, 1, 0.5, S(O=867, T=2735, UT=239, UO=255)),
    ("LeafletMip4", 1877, 4, 1, 0.5, S(O=928, T=2777, UT=199, UO=192)),
    ("LeafletMip5", 1889, 5, 1, 0.5, S(O=965, T=2821, UT=156, UO=154)),
    ("LeafletMip6", 1901, 6, 1, 0.5, S(O=526, T=3335, UT=119, UO=116)),
]
LEAFLET_LEVEL = [
    ("LeafletLevel0", 1913, 0, S(UT=1)),
    ("LeafletLevel1", 1925, 1, S(UT=4)),
    ("LeafletLevel2", 1937, 2, S(T=1, UT=10, UO=5)),
    ("LeafletLevel3", 1949, 3, S(T=16, UT=31, UO=17)),
    ("LeafletLevel4", 1961, 4, S(O=35, T=108, UT=68, UO=45)),
    ("LeafletLevel5", 1973, 5, S(O=207, T=554, UT=139, UO=124)),
    ("LeafletLevel6", 1985, 6, S(O=1021, T=2508, UT=275, UO=292)),
    ("LeafletLevel7", 1997, 7, S(O=4666, T=10580, UT=549, UO=589)),
    ("LeafletLevel8", 2009, 8, S(O=19831, T=43424, UT=1110, UO=1171)),
]
