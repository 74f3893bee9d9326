"""Big-integer checks of the algebra the wrapping circuit relies on (zecale_amd/csrc/circuit/tower.hpp, bls12_377.hpp): pure Python,
no library.  q = BLS12-377 base field = BW6-761 scalar field, r = BLS12-377 group order, u = the curve parameter."""
from zecale_amd.encoding import R_MOD as Q            # BLS12-377 Fq

U = 0x8508c00000000001
R_BLS = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001


def test_curve_parameter_relations():
    assert R_BLS == U**4 - U**2 + 1                                   # r(u)
    assert Q == ((U - 1)**2 * (U**4 - U**2 + 1)) // 3 + U             # q(u) of the BLS12 family


def test_direct_degree_12_extension_is_a_field():
    """Fq12 = Fq[w]/(w^12 + 5): x^12 - a (a = -5) is irreducible over Fq iff a is no p-th power for the primes p | 12 and,
    because 4 | 12, -4a... i.e. a is not of the form -4 b^4 (Lang, Algebra VI.9.1); needs q = 1 mod 12 for the Frobenius constants."""
    a = (-5) % Q
    assert Q % 12 == 1
    assert pow(a, (Q - 1) // 2, Q) != 1                               # not a square
    assert pow(a, (Q - 1) // 3, Q) != 1                               # not a cube
    minus_a_over_4 = 5 * pow(4, -1, Q) % Q                            # a = -4 b^4  <=>  5/4 = b^4
    assert pow(minus_a_over_4, (Q - 1) // 4, Q) != 1                  # 5/4 is not a fourth power
    # u^2 = w^12 = -5: Fq2 = Fq[u]/(u^2 + 5) is the quadratic subfield used by the twist
    assert pow(a, (Q - 1) // 2, Q) == Q - 1


def test_final_exponent_decomposition():
    """final_exponentiation raises to 3 (q^12 - 1)/r = (q^6 - 1)(q^2 + 1) * 3 (q^4 - q^2 + 1)/r with the hard part
    l0 + l1 q + l2 q^2 + l3 q^3,  l3 = (u-1)^2, l2 = l3 u, l1 = l2 u - l3, l0 = l1 u + 3; the factor 3 is coprime to r."""
    l3 = (U - 1)**2
    l2 = l3 * U
    l1 = l2 * U - l3
    l0 = l1 * U + 3
    assert (Q**4 - Q**2 + 1) % R_BLS == 0
    assert l0 + l1 * Q + l2 * Q**2 + l3 * Q**3 == 3 * ((Q**4 - Q**2 + 1) // R_BLS)
    assert (Q**12 - 1) == (Q**6 - 1) * (Q**2 + 1) * (Q**4 - Q**2 + 1)
    assert R_BLS % 3 != 0


def test_line_multiplication_needs_21_points():
    """A line has non-zero coefficients at w^0, w^1, w^3, w^7, w^9: degree 9; times a degree-11 element: degree 20, 21 coefficients."""
    assert max((0, 1, 3, 7, 9)) + 11 + 1 == 21
