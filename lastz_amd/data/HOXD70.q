# HOXD70 -- the substitution scores of Chiaromonte, Yap & Miller (PSB 2002), LASTZ's default matrix
# (src/dna_utilities.c:137-148), as a --scores=<file> the lastz CLI reads (src/dna_utilities.c:562-581,
# src/lastz.c:6115): BASELINE.json configs[4] names it literally (--scores=HOXD70).
bad_score          = X:-1000  # used for sub['X'][*] and sub[*]['X']
fill_score         = -100     # used when sub[*][*] is not otherwise defined
gap_open_penalty   = 400
gap_extend_penalty = 30

      A     C     G     T
A    91  -114   -31  -123
C  -114   100  -125   -31
G   -31  -125   100  -114
T  -123   -31  -114    91
