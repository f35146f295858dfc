{-# LANGUAGE ForeignFunctionInterface #-}
-- | Haskell side of libraytrace_gpu.so: the reference's scene API (src/Main.hs) with `Shape` and
-- `Light` as recorded constructor calls instead of closures, and `parallelTrace` as ONE foreign call.
--
-- WRITTEN BLIND: neither this image nor the GPU box has GHC (probed: ghc, cabal, stack, runghc absent),
-- so this module has never been compiled.  It is the binding a maintainer of the reference would add;
-- the same C entry points are exercised through ctypes (raytrace_amd/_lib.py) and C++
-- (include/raytrace.hpp, examples/main.cpp) by the test suite.  See INTEGRATION.md.
--
-- Usage in src/Main.hs (replaces Main.hs:571 and :621):
--
-- > main = do
-- >   let (world, lights) = linearInterpolation ...          -- built with the constructors below
-- >   img <- parallelTraceGPU 1024 1024 4 lights world
-- >   saveBmpImage "trace.bmp" (ImageRGB8 img)
module Raytrace.GPU
  ( Vector3D(..), Material(..), Shape, Light
  , rectangle, triangle, sphere, cuboid, colorcuboid, cube, colorcube, cylinder
  , gaussianNormals, pointLight, ambient
  , parallelTraceGPU, renderBand
  ) where

import Codec.Picture (Image (..), PixelRGB8 (..))
import Control.Exception (bracket)
import Control.Monad (forM_, void, when)
import Data.Int (Int32)
import qualified Data.Vector.Storable as VS
import Data.Word (Word64, Word8)
import Foreign
import Foreign.C.String (CString, peekCString)
import Foreign.C.Types

-- Main.hs:24
data Vector3D = MkV3D !Double !Double !Double

-- Main.hs:178-185
data Material = MkMaterial
  { diffuse      :: PixelRGB8
  , specular     :: PixelRGB8
  , specularity  :: Double
  , reflectivity :: Double
  }

-- | A recorded shape tree.  `<>` and `mconcat` keep the tree shape, because every mappend node of the
-- reference guards its subtree with a bounding-box test (Main.hs:205-216).
data Shape
  = Rect Material Vector3D Vector3D Vector3D                       -- Main.hs:226
  | Tri Material Vector3D Vector3D Vector3D                        -- Main.hs:363
  | Sph Material Vector3D Double                                   -- Main.hs:392
  | Cuboid [Material] Vector3D Double Double Double                -- Main.hs:301
  | Cyl Material Material Material Vector3D Int Double Double      -- Main.hs:725
  | Append Shape Shape
  | Concat [Shape]
  | GaussNormals Double Vector3D Shape                             -- Main.hs:956-965 with main's fnorm (580-584)

instance Semigroup Shape where
  (<>) = Append
instance Monoid Shape where
  mempty  = Concat []
  mconcat = Concat                                                 -- foldr mappend mempty, same boxes

data LightSrc = Point Double Double Vector3D | Ambient Double
newtype Light = MkLight [LightSrc]
instance Semigroup Light where
  MkLight a <> MkLight b = MkLight (a ++ b)
instance Monoid Light where
  mempty = MkLight []

rectangle :: Material -> Vector3D -> Vector3D -> Vector3D -> Shape
rectangle = Rect
triangle :: Material -> Vector3D -> Vector3D -> Vector3D -> Shape
triangle = Tri
sphere :: Material -> Vector3D -> Double -> Shape
sphere = Sph
colorcuboid :: [Material] -> Vector3D -> Double -> Double -> Double -> Shape
colorcuboid [] _ _ _ _ = error "colorcuboid: list of materials must not be empty."   -- Main.hs:302
colorcuboid ms p l h d = Cuboid ms p l h d
cuboid :: Material -> Vector3D -> Double -> Double -> Double -> Shape
cuboid m = colorcuboid [m]
colorcube :: [Material] -> Vector3D -> Double -> Shape
colorcube ms p l = colorcuboid ms p l l l
cube :: Material -> Vector3D -> Double -> Shape
cube m = colorcube [m]
cylinder :: Material -> Material -> Material -> Vector3D -> Int -> Double -> Double -> Shape
cylinder = Cyl
-- | linearInterpolation's `normals` wrapper for the gaussian fnorm: scaleInv, offset (Main.hs:956-965).
gaussianNormals :: Double -> Vector3D -> Shape -> Shape
gaussianNormals = GaussNormals

pointLight :: Double -> Double -> Vector3D -> Light                 -- Main.hs:467 (the world is bound at render time)
pointLight kd ks p = MkLight [Point kd ks p]
ambient :: Double -> Light                                          -- Main.hs:490
ambient f = MkLight [Ambient f]

-- ---------------------------------------------------------------------------------------------
-- C ABI (include/raytrace_gpu.h)

data RtScene
type ScenePtr = Ptr RtScene

foreign import ccall unsafe "rt_last_error"      c_last_error      :: IO CString
foreign import ccall unsafe "rt_scene_create"    c_scene_create    :: Ptr ScenePtr -> IO CInt
foreign import ccall unsafe "rt_scene_destroy"   c_scene_destroy   :: ScenePtr -> IO ()
foreign import ccall unsafe "rt_scene_num_leaves" c_num_leaves     :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_begin_mconcat"   c_begin_mconcat   :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_end_mconcat"     c_end_mconcat     :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_begin_mappend"   c_begin_mappend   :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_end_mappend"     c_end_mappend     :: ScenePtr -> IO CInt
foreign import ccall unsafe "rt_add_rectangle"   c_add_rectangle   :: ScenePtr -> Ptr Word8 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
foreign import ccall unsafe "rt_add_triangle"    c_add_triangle    :: ScenePtr -> Ptr Word8 -> Ptr CDouble -> Ptr CDouble -> Ptr CDouble -> IO CInt
foreign import ccall unsafe "rt_add_sphere"      c_add_sphere      :: ScenePtr -> Ptr Word8 -> Ptr CDouble -> CDouble -> IO CInt
foreign import ccall unsafe "rt_add_cuboid"      c_add_cuboid      :: ScenePtr -> Ptr Word8 -> CInt -> Ptr CDouble -> CDouble -> CDouble -> CDouble -> IO CInt
foreign import ccall unsafe "rt_add_cylinder"    c_add_cylinder    :: ScenePtr -> Ptr Word8 -> Ptr Word8 -> Ptr Word8 -> Ptr CDouble -> CInt -> CDouble -> CDouble -> IO CInt
foreign import ccall unsafe "rt_set_normal_mode" c_set_normal_mode :: ScenePtr -> CInt -> CInt -> CInt -> Ptr CDouble -> CInt -> IO CInt
foreign import ccall unsafe "rt_add_point_light" c_add_point_light :: ScenePtr -> CDouble -> CDouble -> Ptr CDouble -> IO CInt
foreign import ccall unsafe "rt_add_ambient"     c_add_ambient     :: ScenePtr -> CDouble -> IO CInt
-- long call: `safe`, so other Haskell threads keep running while the GPU renders
foreign import ccall unsafe "rt_host_alloc"      c_host_alloc      :: CSize -> IO (Ptr Word8)
foreign import ccall        "&rt_host_free"      p_host_free       :: FunPtr (Ptr Word8 -> IO ())
foreign import ccall safe   "rt_render"          c_render          :: ScenePtr -> CInt -> CInt -> CInt -> CInt -> Word64 -> CInt -> CInt -> Ptr Word8 -> Ptr Int32 -> IO CInt

check :: String -> CInt -> IO CInt
check what rc = do
  when (rc < 0) $ do
    msg <- c_last_error >>= peekCString
    ioError (userError (what ++ ": " ++ msg))
  return rc

-- rt_material: diffuse[3], specular[3], 2 bytes of padding, specularity, reflectivity (24 bytes, 8-aligned)
withMaterials :: [Material] -> (Ptr Word8 -> IO a) -> IO a
withMaterials ms act = allocaBytesAligned (24 * max 1 (length ms)) 8 $ \p -> do
  forM_ (zip [0 ..] ms) $ \(i, MkMaterial (PixelRGB8 dr dg db) (PixelRGB8 sr sg sb) sp rf) -> do
    let q = p `plusPtr` (24 * i)
    pokeByteOff q 0 dr >> pokeByteOff q 1 dg >> pokeByteOff q 2 db
    pokeByteOff q 3 sr >> pokeByteOff q 4 sg >> pokeByteOff q 5 sb
    pokeByteOff q 6 (0 :: Word8) >> pokeByteOff q 7 (0 :: Word8)
    pokeByteOff q 8 (realToFrac sp :: CDouble)
    pokeByteOff q 16 (realToFrac rf :: CDouble)
  act p

withV3 :: Vector3D -> (Ptr CDouble -> IO a) -> IO a
withV3 (MkV3D x y z) = withArray (map realToFrac [x, y, z])

-- | Replays the recorded tree in depth-first order: the leaf order of the library (and the value in the
-- hit-ID buffer) is the reference's tie order (left-most leaf wins equal t, Main.hs:213-214).
emit :: ScenePtr -> Shape -> IO ()
emit s shape = case shape of
  Append a b -> do
    void $ c_begin_mappend s >>= check "rt_begin_mappend"
    emit s a >> emit s b
    void $ c_end_mappend s >>= check "rt_end_mappend"
  Concat xs -> do
    void $ c_begin_mconcat s >>= check "rt_begin_mconcat"
    mapM_ (emit s) xs
    void $ c_end_mconcat s >>= check "rt_end_mconcat"
  Rect m p w h -> withMaterials [m] $ \pm -> withV3 p $ \pp -> withV3 w $ \pw -> withV3 h $ \ph ->
    void $ c_add_rectangle s pm pp pw ph >>= check "rt_add_rectangle"
  Tri m a b c -> withMaterials [m] $ \pm -> withV3 a $ \pa -> withV3 b $ \pb -> withV3 c $ \pc ->
    void $ c_add_triangle s pm pa pb pc >>= check "rt_add_triangle"
  Sph m c r -> withMaterials [m] $ \pm -> withV3 c $ \pc ->
    void $ c_add_sphere s pm pc (realToFrac r) >>= check "rt_add_sphere"
  Cuboid ms p l h d -> withMaterials ms $ \pm -> withV3 p $ \pp ->
    void $ c_add_cuboid s pm (fromIntegral (length ms)) pp (realToFrac l) (realToFrac h) (realToFrac d) >>= check "rt_add_cuboid"
  Cyl t b m p n h r -> withMaterials [t] $ \pt -> withMaterials [b] $ \pb -> withMaterials [m] $ \pm -> withV3 p $ \pp ->
    void $ c_add_cylinder s pt pb pm pp (fromIntegral n) (realToFrac h) (realToFrac r) >>= check "rt_add_cylinder"
  GaussNormals scaleInv (MkV3D ox oy oz) child -> do
    first <- c_num_leaves s
    emit s child
    n <- c_num_leaves s
    withArray (map realToFrac [scaleInv, ox, oy, oz]) $ \pp ->
      void $ c_set_normal_mode s first (n - first) 2 pp 4 >>= check "rt_set_normal_mode"

emitLight :: ScenePtr -> LightSrc -> IO ()
emitLight s (Point kd ks p) = withV3 p $ \pp -> void $ c_add_point_light s (realToFrac kd) (realToFrac ks) pp >>= check "rt_add_point_light"
emitLight s (Ambient f)     = void $ c_add_ambient s (realToFrac f) >>= check "rt_add_ambient"

withScene :: Light -> Shape -> (ScenePtr -> IO a) -> IO a
withScene (MkLight ls) world act =
  bracket (alloca $ \pp -> c_scene_create pp >>= check "rt_scene_create" >> peek pp) c_scene_destroy $ \s -> do
    emit s world
    mapM_ (emitLight s) ls
    act s

-- | Rows [row0, row0 + nrows) of the w x h image as RGBA8 (row-major, y = 0 is the top row) — the
-- multi-GPU partition: one process per GPU renders one band.
renderBand :: Int -> Int -> Int -> Light -> Shape -> Int -> Int -> IO (ForeignPtr Word8)
renderBand w h depth lights world row0 nrows = withScene lights world $ \s -> do
  -- page-locked memory from the library (freed by rt_host_free when the ForeignPtr dies): the copy back is
  -- DMA behind the render; mallocForeignPtrBytes would work too, through the runtime's bounce buffers
  raw <- c_host_alloc (fromIntegral (max 1 (w * nrows * 4)))
  when (raw == nullPtr) $ c_last_error >>= peekCString >>= \m -> ioError (userError ("rt_host_alloc: " ++ m))
  buf <- newForeignPtr p_host_free raw
  withForeignPtr buf $ \p ->
    void $ c_render s (fromIntegral w) (fromIntegral h) (fromIntegral depth) 1 0 (fromIntegral row0) (fromIntegral nrows) p nullPtr >>= check "rt_render"
  return buf

-- | parallelTrace w h (fixedCamera w h) (trace depth lights world) — Main.hs:623-637 — as one call.
-- The RGBA8 band becomes JuicyPixels' RGB8 storable vector (alpha dropped), so trace.bmp stays 24-bit.
parallelTraceGPU :: Int -> Int -> Int -> Light -> Shape -> IO (Image PixelRGB8)
parallelTraceGPU w h depth lights world = do
  buf <- renderBand w h depth lights world 0 h
  let rgba = VS.unsafeFromForeignPtr0 buf (w * h * 4)
      rgb  = VS.generate (w * h * 3) (\i -> rgba VS.! (4 * (i `div` 3) + i `mod` 3))
  return $! Image w h rgb
