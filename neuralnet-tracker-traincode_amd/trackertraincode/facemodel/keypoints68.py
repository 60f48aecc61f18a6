"""Index sets of the 68-point face annotation (reference: facemodel/keypoints68.py).  Pure data: which
landmark is which, and how left/right swap under a horizontal flip."""

# left/right partner of every landmark under a horizontal mirror
flip_map = (
    list(range(16, -1, -1))          # jaw line 0..16
    + list(range(26, 16, -1))        # brows 17..26
    + [27, 28, 29, 30]               # nose bridge
    + [35, 34, 33, 32, 31]           # nostrils
    + [45, 44, 43, 42, 47, 46]       # right eye <- left eye
    + [39, 38, 37, 36, 41, 40]
    + [54, 53, 52, 51, 50, 49, 48]   # outer lip, upper
    + [59, 58, 57, 56, 55]           # outer lip, lower
    + [64, 63, 62, 61, 60]           # inner lip, upper
    + [67, 66, 65]                   # inner lip, lower
)

# both sides contain the middle point
chin_left = list(range(0, 9))
chin_right = list(range(8, 17))

upperlip_left, upperlip_right = [48, 49, 50, 51], [51, 52, 53, 54]
lowerlip_left, lowerlip_right = [48, 59, 58, 57], [57, 56, 55, 54]
uppermouth_left, uppermouth_right = [60, 61, 62], [62, 63, 64]
lowermouth_left, lowermouth_right = [60, 67, 66], [66, 65, 64]

nose_left, nose_right = [31, 32, 33], [33, 34, 35]
nose_back = [27, 28, 29, 30, 33]

eyecorners_left, eyecorners_right = [36, 39], [42, 45]
brows_left, brows_right = list(range(17, 22)), list(range(22, 27))

eye_left_top, eye_left_bottom = [36, 37, 38, 39], [36, 41, 40, 39]
eye_right_top, eye_right_bottom = [42, 43, 44, 45], [42, 47, 46, 45]
eye_not_corners = [37, 38, 41, 40, 43, 44, 47, 46]

nose_tip = 33
mouth_corner_left = 60
mouth_corner_right = 64
